import subprocess, sys, hashlib, torch, os
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
subprocess.run(["gcc", "-O2", "-o", f"{root}/tools/gen_amplicons", f"{root}/tools/gen_amplicons.c", "-lm"], check=True)
subprocess.run([f"{root}/tools/gen_amplicons", "3000000", "150", "7", "1", "0.3", "/tmp/p.fa"], check=True)
def md5(p): return hashlib.md5(open(p, "rb").read()).hexdigest()[:12]
def run(tag, extra_env=None):
    env = dict(os.environ); env.update(extra_env or {})
    subprocess.run([f"{root}/swarm_amd/bin/swarm", "-d", "1", "-f", "-o", "/tmp/p.o", "-l", "/dev/null", "/tmp/p.fa"], check=True, env=env)
    print(tag, md5("/tmp/p.o"), flush=True)
run("clean")
for pat in (0xFF, 0xAA, 0x01, None):
    t = torch.empty(220 * (1 << 30), dtype=torch.uint8, device="cuda")
    if pat is None:
        t.random_(0, 256)
    else:
        t.fill_(pat)
    torch.cuda.synchronize()
    del t
    torch.cuda.empty_cache()
    run(f"poison {pat}")
    run(f"poison {pat} plain", {"SWA_D1_PLAIN": "1"})
subprocess.run([f"{root}/oracle/_ref/swarm", "-d", "1", "-f", "-t", "32", "-o", "/tmp/pr.o", "-l", "/dev/null", "/tmp/p.fa"], check=True)
print("reference", md5("/tmp/pr.o"))
