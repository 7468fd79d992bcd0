"""A stock-PyTorch workload of the hot path's shape (no kernel of this repository): 7 bf16 GEMMs of 153,600 x 512 x 512 forward and backward,
a 5-level gather and an atomic scatter into 420 MB of maps, AdamW -- eagerly and as a captured graph.  For telling a faulty GPU box from a
faulty kernel: tools/stress_trees.sh's intermittent HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION (round 6) against this on the same box."""
import sys
import torch

dev = torch.device("cuda", 0)
torch.manual_seed(0)
M, H = 153600, 512
net = torch.nn.Sequential(*[m for _ in range(7) for m in (torch.nn.Linear(H, H), torch.nn.ReLU())]).to(dev).to(torch.bfloat16)
opt = torch.optim.AdamW(net.parameters(), lr=1e-4, capturable=True, fused=True)
maps = torch.randn(1500 * 452, 64, device=dev).requires_grad_(True)
idx = torch.randint(0, 1500 * 452, (M, 4), device=dev)
w = torch.rand(M, 4, 1, device=dev)


def step():
    opt.zero_grad(set_to_none=True)
    maps.grad = None
    z = (maps[idx] * w).sum(1)                    # gather (backward: atomic scatter)
    x = torch.cat([z] * 8, dim=1).to(torch.bfloat16)
    y = net(x)
    loss = y.float().square().mean()
    loss.backward()
    opt.step()
    return loss


steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
for _ in range(3):
    step()
torch.cuda.synchronize()
if len(sys.argv) > 2 and sys.argv[2] == "graph":
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        last = step()
    for _ in range(steps):
        g.replay()
else:
    for _ in range(steps):
        last = step()
torch.cuda.synchronize()
print("ok", float(last))
