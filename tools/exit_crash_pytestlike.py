"""Development aid (round 4): examples/strong_hip started the way the test suite starts it -- from a python process that holds its
own HIP context on the same GPU (torch), stdout/stderr into a pipe -- N times; counts the runs that end non-zero AFTER printing
their result.  With --gdb every run goes through rocgdb and the first dying one prints its backtraces."""
import subprocess, sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
gdb = "--gdb" in sys.argv
args = ["64", "32", "8"]
x = torch.zeros(1 << 20, device="cuda")      # the parent owns a context, as pytest does
torch.cuda.synchronize()
exe = os.path.join(ROOT, "examples", "strong_hip")
bad = []
for i in range(n):
    if gdb:
        cmd = ["/opt/rocm/bin/rocgdb", "-q", "-batch", "-ex", "set pagination off", "-ex", "run", "-ex", "bt 30", "-ex", "thread apply all bt 12", "--args", exe, *args]
    else:
        cmd = [exe, *args]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    out = r.stdout.decode(errors="replace")
    died = ("SIGSEGV" in out or "SIGABRT" in out) if gdb else r.returncode != 0
    if died:
        bad.append((i, r.returncode))
        keep = [l for l in out.splitlines() if not l.startswith(("[New Thread", "[Thread", "rank "))]
        print(f"== run {i} died (rc {r.returncode})")
        print("\n".join(keep[-120:]))
        if gdb:
            break
    if i % 10 == 9:
        print(f"{i + 1} runs, {len(bad)} died", flush=True)
print(f"{len(bad)} of {i + 1} runs died: {bad}")
