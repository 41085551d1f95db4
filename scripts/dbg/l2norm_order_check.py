import sys, numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, 'ransac-flow_amd')
from rfx import ops
dev = torch.device('cuda:0')
def fma(a, b, c): return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)
print(torch.backends.cpu.get_cpu_capability(), torch.get_num_threads())
for shape in ((2, 1024, 30, 40), (1, 256, 60, 80), (1, 50, 9, 11)):
    g = torch.Generator().manual_seed(shape[1] + shape[2])
    x = torch.relu(torch.randn(*shape, generator=g)) * torch.rand(1, shape[1], 1, 1, generator=g)
    want = F.normalize(x)
    got = ops.l2norm(x.to(dev)).cpu()
    N, C = shape[0], shape[1]
    X = x.reshape(N, C, -1).numpy()
    s = np.zeros((N, X.shape[2]), np.float32)
    for c in range(C): s = fma(X[:, c], X[:, c], s)
    d = np.maximum(np.sqrt(s).astype(np.float32), np.float32(1e-12))
    emu = torch.from_numpy((X / d[:, None, :]).astype(np.float32)).reshape(shape)
    tn = x.norm(2, dim=1).reshape(N, -1).numpy()
    print(shape, "device==torch %.5f  device==emu %.5f  torch==emu %.5f  torch.norm==emu norm %.5f  max|dev-torch| %.2e" % (
        (got == want).float().mean(), (got == emu).float().mean(), (want == emu).float().mean(), (tn == np.sqrt(s).astype(np.float32)).mean(), (got - want).abs().max()))
    for th in (1, 8):
        torch.set_num_threads(th)
        print("   threads", th, "torch==emu %.5f" % (F.normalize(x) == emu).float().mean())
    torch.set_num_threads(64)
