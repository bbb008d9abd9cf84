import sys, time, torch
sys.path.insert(0, "/root/repo")
from difffacto_amd.pointnet2_ops import pointnet2_modules as pm
torch.manual_seed(0)
for spec, B, M, ns in (([6, 64, 64, 128], 16, 512, 32), ([131, 128, 128, 256], 16, 128, 64), ([259, 256, 512, 1024], 16, 1, 128)):
    seq = pm.build_shared_mlp(list(spec), bn=True).cuda().train()
    x = torch.randn(B, spec[0], M, ns, device="cuda", requires_grad=True)
    def run(native):
        y = pm.shared_mlp_train(seq, x, pool=True) if native else seq(x).amax(dim=3)
        y.sum().backward()
    for native in (False, True):
        for _ in range(3): run(native)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): run(native)
        torch.cuda.synchronize()
        print(f"{spec} rows {B*M*ns}: {'libdfx' if native else 'torch '} fwd+bwd {(time.perf_counter()-t0)/10*1e3:.3f} ms")
