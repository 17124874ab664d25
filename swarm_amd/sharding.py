"""Multi-GPU sharding of the d=1 path: the database, hash table and Bloom
filter are replicated on every GPU, the QUERY range [0, n) is cut into contiguous slices (one
per rank), every rank produces the CSR of its slice, and the slices are exchanged with
all-gather (RCCL over xGMI on GPUs — backend "nccl" IS RCCL on ROCm —, gloo in the CPU tests).

Everything here is backend-agnostic torch.distributed plumbing on tensors; the per-rank CSR
comes from swa_d1_network_device on a GPU (bench.py) or from whatever the caller supplies.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


def partition_even(n: int, world: int) -> list:
    """Contiguous (first, count) slices with counts differing by at most one."""
    base, extra = divmod(n, world)
    out, first = [], 0
    for r in range(world):
        count = base + (1 if r < extra else 0)
        out.append((first, count))
        first += count
    return out


def partition_by_length(seqlen: np.ndarray, world: int) -> list:
    """Contiguous slices balanced by total nucleotides (work per query ~ 7 L probes)."""
    n = int(seqlen.shape[0])
    if n == 0:
        return [(0, 0)] * world
    csum = np.cumsum(seqlen.astype(np.int64))
    total = int(csum[-1])
    cuts = [0]
    for r in range(1, world):
        target = total * r // world
        cuts.append(int(np.searchsorted(csum, target, side="left")))
    cuts.append(n)
    for i in range(1, len(cuts)):
        cuts[i] = max(cuts[i], cuts[i - 1])
    return [(cuts[r], cuts[r + 1] - cuts[r]) for r in range(world)]


def allgather_csr(local_offsets: torch.Tensor, local_nb: torch.Tensor, local_total: int, counts: list,
                  group=None):
    """Exchange per-rank CSR slices so that every rank holds the whole network.

    local_offsets : int64 [count_r + 1]  row offsets of this rank's slice (starting at 0)
    local_nb      : int32 [>= local_total] neighbour ids of this rank's slice
    counts        : per-rank query counts (from partition_*), identical on all ranks
    Returns (offsets int64 [n + 1], neighbours int32 [total]) on the tensors' device.

    Collectives: one all_gather of the hit counts and longest rows (16 bytes per rank), one of the
    per-row link counts (1 byte per amplicon, 4 if some row has more than 255 links — not the 64-bit
    offsets: they are a prefix sum away) padded to the largest slice, one of the hit lists padded to
    the largest hit count.
    """
    world = dist.get_world_size(group)
    dev = local_offsets.device
    rows_here = local_offsets.numel() - 1
    row_cnt = local_offsets[1:] - local_offsets[:-1]
    longest = row_cnt.max() if rows_here > 0 else torch.zeros((), dtype=torch.int64, device=dev)
    mine = torch.stack([torch.tensor(local_total, dtype=torch.int64, device=dev), longest.to(torch.int64)])
    gathered = torch.empty(2 * world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(gathered, mine, group=group)
    gathered_host = [int(x) for x in gathered.tolist()]
    totals_host, longest_row = gathered_host[0::2], max(gathered_host[1::2])
    max_rows = max(max(counts), 1)
    max_hits = max(max(totals_host), 1)

    # per-row link counts travel as bytes when every row of the network has at most 255 links (an amplicon has about
    # two neighbours; a quarter of the traffic of 32-bit counts), else as 32-bit integers
    cnt_dtype = torch.uint8 if longest_row <= 255 else torch.int32
    pad_cnt = torch.zeros(max_rows, dtype=cnt_dtype, device=dev)
    pad_cnt[:rows_here] = row_cnt.to(cnt_dtype)
    all_cnt = torch.empty(world * max_rows, dtype=cnt_dtype, device=dev)
    dist.all_gather_into_tensor(all_cnt, pad_cnt, group=group)

    pad_nb = torch.zeros(max_hits, dtype=torch.int32, device=dev)
    pad_nb[:local_total] = local_nb[:local_total]
    all_nb = torch.empty(world * max_hits, dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(all_nb, pad_nb, group=group)

    n = sum(counts)
    if all(c == max_rows for c in counts):
        row_counts = all_cnt                                   # equal slices: already contiguous
    else:
        row_counts = torch.cat([all_cnt[r * max_rows: r * max_rows + counts[r]] for r in range(world)])
    offsets = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(row_counts, dim=0, dtype=torch.int64, out=offsets[1:])
    pieces = [all_nb[r * max_hits: r * max_hits + totals_host[r]] for r in range(world)]
    neighbours = torch.cat(pieces) if pieces else torch.empty(0, dtype=torch.int32, device=dev)
    return offsets, neighbours


def exchange_owned_links(links: torch.Tensor, counts: list, group=None):
    """Ownership sharding (swa_d1_set_ownership): every rank holds the links it found in the anchor
    groups it owns, as one flat list (swa_d1_network_edges_device), and every link of the network
    is held by exactly one rank.  The links are redistributed by contiguous seed range and turned
    into this rank's slice of the CSR.  Nothing here is proportional to the database size.

    links  : int64 [m]  source << 32 | target (ids < 2^31), any order
    counts : per-rank row counts of the final partition (from partition_*), sum = n
    Returns (offsets int64 [counts[rank] + 1] starting at 0, neighbours int32, rows ascending) on
    the links' device — what allgather_csr takes.

    Collectives: all-to-all of the split sizes (8 bytes per pair of ranks), all-to-all of the
    links (8 bytes each); no link travels twice.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = links.device
    bounds = [0]
    for c in counts:
        bounds.append(bounds[-1] + c)
    assert bounds[-1] < (1 << 31)
    mine, first = counts[rank], bounds[rank]

    keys, _ = torch.sort(links)                               # by source: destinations become contiguous
    cuts = torch.searchsorted(keys, torch.tensor(bounds, dtype=torch.int64, device=dev) << 32)
    send_sizes = cuts[1:] - cuts[:-1]
    recv_sizes = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_to_all_single(recv_sizes, send_sizes, group=group)
    send_list = [int(x) for x in send_sizes.tolist()]
    recv_list = [int(x) for x in recv_sizes.tolist()]
    got = torch.empty(sum(recv_list), dtype=torch.int64, device=dev)
    dist.all_to_all_single(got, keys, output_split_sizes=recv_list, input_split_sizes=send_list, group=group)

    got, _ = torch.sort(got)                                  # rows ascending, links ascending within a row
    rows = (got >> 32) - first
    offsets = torch.zeros(mine + 1, dtype=torch.int64, device=dev)
    if got.numel() > 0:
        torch.cumsum(torch.bincount(rows, minlength=mine), dim=0, out=offsets[1:])
    neighbours = (got & 0xFFFFFFFF).to(torch.int32)
    return offsets, neighbours


def exchange_routed_ids(d_ids: torch.Tensor, d_counts: torch.Tensor, cap: int, group=None):
    """Routed index build, step 2: what swa_d1_route_slice left on every rank — for each anchor index (0 prefix side,
    1 suffix side) and each owning rank the ids of the rank's own slice, d_ids[(index * world + owner) * cap + ..],
    d_counts[index * world + owner] — travels all-to-all; every rank gets back the members of the groups it owns.

    Returns (ids_prefix int32 [m0], ids_suffix int32 [m1]) on the tensors' device: the arguments of
    swa_d1_index_build_routed.  Collectives: all-to-all of the 2 counts per pair of ranks, all-to-all of the ids
    (4 bytes per amplicon and index in total: nothing here is proportional to the database on any one rank)."""
    world = dist.get_world_size(group)
    dev = d_ids.device
    assert int(d_counts[2 * world]) == 0, "swa_d1_route_slice: a destination region overflowed"
    send = d_counts[:2 * world].to(torch.int64).view(2, world).t().contiguous()          # [owner, index]
    recv = torch.empty_like(send)                                                          # [source, index]
    dist.all_to_all_single(recv, send, group=group)
    send_h, recv_h = send.tolist(), recv.tolist()
    regions = d_ids.view(2 * world, cap)
    out_buf = torch.cat([regions[index * world + owner, :send_h[owner][index]] for owner in range(world) for index in range(2)])
    got = torch.empty(sum(a + b for a, b in recv_h), dtype=d_ids.dtype, device=dev)
    dist.all_to_all_single(got, out_buf, output_split_sizes=[a + b for a, b in recv_h], input_split_sizes=[a + b for a, b in send_h],
                           group=group)
    pieces, at = ([], []), 0
    for a, b in recv_h:
        pieces[0].append(got[at: at + a]); pieces[1].append(got[at + a: at + a + b])
        at += a + b
    return torch.cat(pieces[0]).contiguous(), torch.cat(pieces[1]).contiguous()


def exchange_routed_records(d_rec: torch.Tensor, d_counts: torch.Tensor, cap: int, group=None):
    """Routed index build with the KEY RECORDS travelling, step 2: what swa_d1_route_slice_records left on every rank — per
    anchor index and owning rank the finished key records of the rank's own slice, d_rec[(index * world + owner) * cap + ..]
    — travels all-to-all; every rank gets back the records of the groups it owns and starts its build at the partition (no
    keying pass over received ids).

    Returns (records_prefix int64 [m0], records_suffix int64 [m1]): the arguments of swa_d1_index_build_records.
    Collectives: all-to-all of the 2 counts per pair of ranks, ONE all-to-all of the records of both indexes (8 bytes each:
    16 bytes per amplicon in all)."""
    world = dist.get_world_size(group)
    dev = d_rec.device
    assert int(d_counts[2 * world]) == 0, "swa_d1_route_slice_records: a destination region overflowed"
    send = d_counts[:2 * world].to(torch.int64).view(2, world).t().contiguous()          # [owner, index]
    recv = torch.empty_like(send)                                                          # [source, index]
    dist.all_to_all_single(recv, send, group=group)
    send_h, recv_h = send.tolist(), recv.tolist()
    regions = d_rec.view(2 * world, cap)
    out_rec = torch.cat([regions[index * world + owner, :send_h[owner][index]] for owner in range(world) for index in range(2)])
    got = torch.empty(sum(a + b for a, b in recv_h), dtype=d_rec.dtype, device=dev)
    dist.all_to_all_single(got, out_rec, output_split_sizes=[a + b for a, b in recv_h], input_split_sizes=[a + b for a, b in send_h],
                           group=group)
    pieces, at = ([], []), 0
    for a, b in recv_h:
        pieces[0].append(got[at: at + a]); pieces[1].append(got[at + a: at + a + b])
        at += a + b
    return torch.cat(pieces[0]).contiguous(), torch.cat(pieces[1]).contiguous()


def combine_grafts(graft_cand, counters, device=None, group=None):
    """Fastidious pass split over ranks by heavy-amplicon slice (swa_d1_fastidious_shard):
    the reference keeps, per light amplicon, the smallest heavy id that reaches it
    (src/algod1.cc:244-258), so the shards combine with an element-wise MIN all-reduce
    (SWA_NO_AMPLICON = 0xFFFFFFFF is the neutral element); "heavy variants" and "graft
    candidates" (counters[1], [2]) add up, the other counters are the same on every rank.

    graft_cand : uint32 [n] numpy array (this rank's shard result)
    counters   : uint64 [>= 5] numpy array
    Returns (graft_cand uint32 [n], counters uint64) as numpy arrays, identical on all ranks.
    """
    g = torch.from_numpy(np.ascontiguousarray(graft_cand, dtype=np.uint32).astype(np.int64))
    c = torch.from_numpy(np.ascontiguousarray(counters, dtype=np.uint64).astype(np.int64))
    add = c[1:3].clone()
    if device is not None:
        g, add = g.to(device), add.to(device)
    dist.all_reduce(g, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(add, op=dist.ReduceOp.SUM, group=group)
    out_c = np.array(counters, dtype=np.uint64, copy=True)
    out_c[1:3] = add.cpu().numpy().astype(np.uint64)
    return g.cpu().numpy().astype(np.uint32), out_c


def merge_sorted_lists(lists: list, payloads: list):
    """k-way merge of sorted int64 key lists whose keys are distinct across lists (what rank 0 of a multi-GPU d >= 2 job
    does with the ranks' shares — multi.hip: k_merge_sorted_lists): element i of list r goes to i + the number of smaller
    keys in every other list.  Returns (keys, payload) merged; no sort of the whole."""
    total = sum(int(k.numel()) for k in lists)
    dev = lists[0].device
    out_k = torch.empty(total, dtype=torch.int64, device=dev)
    out_p = torch.empty(total, dtype=payloads[0].dtype, device=dev)
    for r, (k, p) in enumerate(zip(lists, payloads)):
        if k.numel() == 0:
            continue
        pos = torch.arange(k.numel(), dtype=torch.int64, device=dev)
        for s, other in enumerate(lists):
            if s != r and other.numel() != 0:
                pos += torch.searchsorted(other, k, right=s < r)
        out_k[pos] = k
        out_p[pos] = p
    return out_k, out_p


def gather_dn_graph(keys: torch.Tensor, diffs: torch.Tensor, n: int, group=None):
    """d >= 2 on several GPUs (second paragraph; the reference's fan-out: src/scan.cc:221-256 under the loop of
    src/algo.cc:505-602).  Every rank holds the pairs of the window groups it owns (swa_dn_set_ownership) as a SORTED list
    of triples — keys = query << 32 | target, diffs = differences of the pair — and every pair of the graph is held by
    exactly one rank.  The lists travel to rank 0 (collectives: one all_gather of the counts, one gather of the keys, one
    of the diffs, both padded to the longest list), which merges them (merge_sorted_lists) into the CSR the greedy walk
    runs over.

    Returns on rank 0 (offsets int64 [n + 1], neighbours int32 [total], diffs uint8 [total]); None elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = keys.device
    mine = torch.tensor([keys.numel()], dtype=torch.int64, device=dev)
    counts = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, mine, group=group)
    counts_h = [int(c) for c in counts.tolist()]
    longest = max(max(counts_h), 1)
    pad_k = torch.zeros(longest, dtype=torch.int64, device=dev)
    pad_k[:keys.numel()] = keys
    pad_d = torch.zeros(longest, dtype=torch.uint8, device=dev)
    pad_d[:diffs.numel()] = diffs.to(torch.uint8)
    got_k = [torch.empty(longest, dtype=torch.int64, device=dev) for _ in range(world)] if rank == 0 else None
    got_d = [torch.empty(longest, dtype=torch.uint8, device=dev) for _ in range(world)] if rank == 0 else None
    dist.gather(pad_k, got_k, dst=0, group=group)
    dist.gather(pad_d, got_d, dst=0, group=group)
    if rank != 0:
        return None
    lists = [got_k[r][:counts_h[r]] for r in range(world)]
    pays = [got_d[r][:counts_h[r]] for r in range(world)]
    merged, mdiff = merge_sorted_lists(lists, pays)
    bounds = torch.arange(n + 1, dtype=torch.int64, device=dev) << 32
    offsets = torch.searchsorted(merged, bounds)
    return offsets, (merged & 0xFFFFFFFF).to(torch.int32), mdiff
