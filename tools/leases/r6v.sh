#!/bin/bash
# lease r6v: tiny and odd inputs through the command line against the reference binary (the prepared / pinned result path, the narrow sort's bit widths)
T=/tmp/r6v; mkdir -p $T; cd $T
S="ACGTTGCAAGCTTAGCGATCGGATCCATGCAAGTCTAGCTAGGCTAACGTACGATCGATCGTAGCTAGCTAGCATCGATCAGCTACGACTAGCATCAGCTACGGATCGATTACGATCAGCTAGCATCGACTGACTAGCTACGATC"
printf ">a_5\n$S\n" > one.fa
printf ">a_5\n$S\n>b_3\n${S:0:100}TTTTT${S:105}\n" > two_unrelated.fa
printf ">a_5\n$S\n>b_3\n${S:0:100}T${S:101}\n>c_1\n${S:0:50}${S:51}\n" > three_linked.fa
printf ">a_5\n$S\n>b_3\n$S\n>c_1\n${S:0:50}${S:51}\n" > with_twins.fa
python3 - <<'PY'
import random
random.seed(3)
S="".join(random.choice("ACGT") for _ in range(150))
with open("singletons.fa","w") as f:
    for i in range(3000):
        s="".join(random.choice("ACGT") for _ in range(random.randint(140,160)))
        f.write(f">s{i}_{random.randint(1,9)}\n{s}\n")
with open("one_big_star.fa","w") as f:
    f.write(f">c_1000\n{S}\n")
    k=0
    for p in range(150):
        for b in "ACGT":
            if S[p]!=b:
                f.write(f">v{k}_1\n{S[:p]+b+S[p+1:]}\n"); k+=1
PY
fail=0
for f in one two_unrelated three_linked singletons one_big_star; do
  for extra in "" "-f" "-n"; do
    rm -f r.o r.s r.i r.w g.o g.s g.i g.w
    $GRAFT_REPO_ROOT/oracle/_ref/swarm -d 1 $extra -o r.o -s r.s -i r.i -w r.w -l /dev/null $f.fa 2>/dev/null || { echo "REF FAIL $f $extra"; fail=1; }
    $GRAFT_REPO_ROOT/swarm_amd/bin/swarm -d 1 $extra -o g.o -s g.s -i g.i -w g.w -l /dev/null $f.fa 2> g.err || { echo "FAIL run $f $extra"; cat g.err; fail=1; }
    for k in o s i w; do cmp -s r.$k g.$k || { echo "DIFF $f '$extra' $k"; fail=1; }; done
  done
done
$GRAFT_REPO_ROOT/oracle/_ref/swarm -d 1 -o r.o -l /dev/null with_twins.fa 2> r.err; rr=$?; $GRAFT_REPO_ROOT/swarm_amd/bin/swarm -d 1 -o g.o -l /dev/null with_twins.fa 2> g.err; gr=$?
echo "twins: ref rc=$rr ours rc=$gr"; cmp r.err g.err && echo "twins: same message"
touch empty.fa
$GRAFT_REPO_ROOT/oracle/_ref/swarm -d 1 -o r.o -l /dev/null empty.fa; echo "ref empty rc=$?"; $GRAFT_REPO_ROOT/swarm_amd/bin/swarm -d 1 -o g.o -l /dev/null empty.fa; echo "ours empty rc=$?"; cmp r.o g.o && echo "empty equal"
echo "fail=$fail"
