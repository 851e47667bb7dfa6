"""the small GEMM of the randomized layer's backward, grad_W = (S G)^T (S X) with K = p: hipBLASLt as called, against the same
product split along K into a batched GEMM + a sum (more workgroups for a 768 x 768 output)"""
import torch

def timed(f, reps=50):
    for _ in range(10):
        f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps):
        f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

for dtype in (torch.bfloat16, torch.float32):
    for p, fout, fin in ((3276, 768, 768), (3276, 3072, 768), (3276, 768, 3072)):
        proj = torch.randn(p, fout, device='cuda', dtype=dtype)
        sk = torch.randn(p, fin, device='cuda', dtype=dtype)
        base = timed(lambda: proj.T @ sk)
        row = [f'plain {base:.1f}']
        want = (proj.T.float() @ sk.float())
        for z in (2, 3, 4, 6, 12):
            if p % z:
                continue
            f = lambda: torch.bmm(proj.view(z, p // z, fout).transpose(1, 2), sk.view(z, p // z, fin)).sum(0)
            err = float((f().float() - want).abs().max() / want.abs().max())
            row.append(f'z={z} {timed(f):.1f} (err {err:.1e})')
        flops = 2 * p * fout * fin
        print(f'{str(dtype)[6:]:9s} p={p} {fout}x{fin}: ' + ' | '.join(row) + f'   [{flops / base / 1e6:.0f} TFLOP/s plain]', flush=True)
