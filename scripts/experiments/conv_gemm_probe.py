"""Probe: 3x3 stride-1 convolution of [B,128,6,6] activations as im2col + one rocBLAS sgemm (NHWC) vs MIOpen's conv2d."""
import time, torch, torch.nn.functional as F
B, C, H, W = 512, 128, 6, 6
x = torch.randn(B, C, H, W, device='cuda'); w = torch.randn(C, C, 3, 3, device='cuda') * 0.03
def t(fn, n=30):
  for _ in range(3): fn()
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(n): fn()
  torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
ref = F.conv2d(x, w, padding=1)
print('conv2d        %.1f us' % (1e6 * t(lambda: F.conv2d(x, w, padding=1))))
# NHWC im2col: pad, 9 shifted views, cat on channel -> [B*H*W, 9C] @ [9C, Cout]
xn = x.permute(0, 2, 3, 1).contiguous()
w2 = w.permute(2, 3, 1, 0).reshape(9 * C, C).contiguous()          # [(ky,kx,cin), cout]
def gemm_nhwc(xn):
  xp = F.pad(xn, (0, 0, 1, 1, 1, 1))
  cols = torch.cat([xp[:, ky:ky + H, kx:kx + W, :] for ky in range(3) for kx in range(3)], dim=3)
  return (cols.view(B * H * W, 9 * C) @ w2).view(B, H, W, C)
out = gemm_nhwc(xn)
print('max diff', float((out.permute(0, 3, 1, 2) - ref).abs().max()))
print('im2col+gemm   %.1f us' % (1e6 * t(lambda: gemm_nhwc(xn))))
cols = torch.cat([F.pad(xn, (0, 0, 1, 1, 1, 1))[:, ky:ky + H, kx:kx + W, :] for ky in range(3) for kx in range(3)], dim=3).view(B * H * W, 9 * C)
print('  gemm alone  %.1f us' % (1e6 * t(lambda: cols @ w2)))
# unfold (NCHW) + matmul
w3 = w.view(C, C * 9)
def gemm_unfold(x):
  return (w3 @ F.unfold(x, 3, padding=1)).view(B, C, H, W)
print('max diff', float((gemm_unfold(x) - ref).abs().max()))
print('unfold+bmm    %.1f us' % (1e6 * t(lambda: gemm_unfold(x))))
