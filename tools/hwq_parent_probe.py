import os, subprocess, sys, torch
n = int(sys.argv[1])
streams = [torch.cuda.Stream() for _ in range(n)]
x = torch.zeros(1024, device="cuda")
for s in streams:
    with torch.cuda.stream(s):
        x += 1
torch.cuda.synchronize()
env = dict(os.environ, GPU_MAX_HW_QUEUES="16")
r = subprocess.run([sys.executable, "tools/dropin_throughput.py", "--threads", "64,256", "--seconds", "3", "--rgb", "--libheif", "libheif_hipcolor.so"], env=env, capture_output=True, text=True)
print("parent GPU_MAX_HW_QUEUES=%s, %d parent streams used:" % (os.environ.get("GPU_MAX_HW_QUEUES"), n))
print("\n".join(l[:120] for l in r.stdout.strip().splitlines()[-2:]))
