import torch, sys
dev = torch.device("cuda:0")
x = torch.ones(1 << 16, device=dev)
bufs = [torch.zeros(1 << 16, device=dev) for _ in range(8)]
s = torch.cuda.Stream()
def body():
    for r in range(40):
        for i in range(8):
            torch.add(x, float(i), out=bufs[i])
            bufs[i].mul_(2.0)
g = torch.cuda.CUDAGraph()
torch.cuda.synchronize()
with torch.cuda.stream(s):
    with torch.cuda.graph(g, stream=s):
        body()
    g.replay()
torch.cuda.synchronize()
ref = [float(b.sum()) for b in bufs]
print("ref", ref[:3])
mode = sys.argv[1] if len(sys.argv) > 1 else "BST"
junk = []
if "B" in mode: junk += [torch.full((1 << 20,), float("nan"), device=dev) for _ in range(2000)]
if "S" in mode: junk += [torch.full((1 << 12,), float("nan"), device=dev) for _ in range(20000)]
if "T" in mode: junk += [torch.full((16,), float("nan"), device=dev) for _ in range(20000)]
torch.cuda.synchronize()
for b in bufs: b.zero_()
with torch.cuda.stream(s):
    g.replay()
torch.cuda.synchronize()
got = [float(b.sum()) for b in bufs]
print("after junk", got[:3], "SAME" if got == ref else "DIFFERENT")
