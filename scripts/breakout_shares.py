#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --stats of `bench.py --workload breakout` -> the kernel-level view of that line: the dominant
kernel with its average duration and share, and the share of GPU time by category (MIOpen / rocBLAS convolutions and GEMMs,
BatchNorm + elementwise, gathers / copies, the two tree kernels of the engine).
usage: breakout_shares.py <kernel_stats.csv> <out.json>"""
import csv, json, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
CATS = [
    ('tree kernels (mz_select / mz_expand_backup / root / finalize)', r'k_tree_|k_dirichlet|k_env_step|k_gather_hidden|k_scatter_hidden|k_store'),
    ('BatchNorm + skip + ReLU epilogues (mz_affine_relu)', r'k_affine_relu'),
    ('convolution (MIOpen)', r'[Cc]onv|igemm|Igemm|gfx9.*_fwd|miopen.*(Fwd|fwd)|naive_conv|Winograd|winograd|sp3|Sp3|ConvBin|gridwise_convolution|implicit_gemm'),
    ('GEMM (rocBLAS / hipBLASLt: the fully connected heads)', r'Cijk_|gemm|Gemm|GEMM'),
    ('copies / gathers (index_select, hidden pool, records)', r'copyBuffer|index_select|indexSelect|gather|CatArrayBatchedCopy|copy_kernel|direct_copy|fillBuffer|memcpy|Memcpy'),
    ('BatchNorm / elementwise / reductions (PyTorch)', r'elementwise|vectorized|reduce|Reduce|softmax|SoftMax|batch_norm|BatchNorm|layer_norm|addcmul|unrolled|where|clamp|random|distribution|philox'),
]
cats = {name: dict(ns=0.0, kernels=0, calls=0) for name, _ in CATS}
cats['other'] = dict(ns=0.0, kernels=0, calls=0)
assigned = {}
for r in rows:
  name = r['Name']
  for cname, pat in CATS:
    if re.search(pat, name):
      break
  else:
    cname = 'other'
  cats[cname]['ns'] += float(r['TotalDurationNs']); cats[cname]['kernels'] += 1; cats[cname]['calls'] += int(r['Calls'])
  assigned[name] = cname
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
top = rows[0]
out = {
    'source': sys.argv[1].split('gpurun_out/')[-1],
    'gpu_time_ms': tot / 1e6,
    'dominant_kernel': {'name': top['Name'][:200], 'calls': int(top['Calls']), 'avg_us': float(top['AverageNs']) / 1e3,
                        'share': float(top['TotalDurationNs']) / tot, 'category': assigned[top['Name']]},
    'top10': [{'name': r['Name'][:120], 'calls': int(r['Calls']), 'avg_us': float(r['AverageNs']) / 1e3,
               'share': float(r['TotalDurationNs']) / tot, 'category': assigned[r['Name']]} for r in rows[:10]],
    'share_by_category': {k: {'share': v['ns'] / tot, 'distinct_kernels': v['kernels'], 'calls': v['calls']} for k, v in cats.items()},
}
json.dump(out, open(sys.argv[2], 'w'), indent=1)
print(json.dumps(out['dominant_kernel']))
print({k: round(v['share'], 4) for k, v in out['share_by_category'].items()})
