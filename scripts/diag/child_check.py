import subprocess, sys, torch
torch.cuda.init(); x = torch.zeros(4, device="cuda"); torch.cuda.synchronize()
print("parent gpu ok", flush=True)
r = subprocess.run([sys.executable, "-c", "print('child exec ok')"], capture_output=True, text=True)
print("subprocess rc", r.returncode, r.stdout.strip(), r.stderr.strip()[-300:], flush=True)
import torch.multiprocessing as mp
def w(rank):
    import torch
    y = torch.ones(4, device="cuda"); torch.cuda.synchronize(); print("spawned child", rank, "gpu ok", float(y.sum()), flush=True)
if __name__ == "__main__":
    mp.spawn(w, nprocs=2, join=True)
    print("mp.spawn ok", flush=True)
