for rep in 1 2; do for pr in 0 -1; do
SHADOW_PREFETCH_PRIORITY=$pr timeout 300 python bench.py --no-cpu-baseline --no-tail 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('prio=$pr', d['ms_per_step'], d['roofline']['frac'], 'sampler', k['sg_sample_pipeline']['avg_ms'], 'reloc', k['sg_relocate_kernel']['avg_ms'], 'gather', k['gather_F100']['avg_ms'], 'spmm100', k['spmm_F100']['avg_ms'], 'nt', k['gemm_nt_split_N256']['avg_ms'], 'Ktail', k['gemm_nt_split_N256_Ktail']['avg_ms'])"
done; done
