#!/usr/bin/env python3
"""Per-kernel register / instruction-mix summary of a hipcc -save-temps .s file (development tool)."""
import re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ''
for m in re.finditer(r'^(_ZN3sgr\w+):.*?\.end_amdhsa_kernel', s, re.S | re.M):
    name = m.group(1)
    if pat not in name:
        continue
    body = m.group(0)
    f = lambda k: re.search(k + r'\s+(\d+)', body).group(1)
    cnt = lambda p: len(re.findall(p, body))
    print(name, 'vgpr', f(r'\.amdhsa_next_free_vgpr'), 'sgpr', f(r'\.amdhsa_next_free_sgpr'), 'lds', f(r'\.amdhsa_group_segment_fixed_size'),
          'scratch', f(r'\.amdhsa_private_segment_fixed_size'), '| pk_fma', cnt(r'v_pk_fma_f32'), 'pk_mul', cnt('v_pk_mul_f32'),
          'pk_add', cnt('v_pk_add_f32'), 'exp', cnt('v_exp_f32'), 'rcp/rsq', cnt(r'v_rcp_f32|v_rsq_f32'), 'fma', cnt(r'v_fma_f32|v_fmac_f32'),
          'mul', cnt(r'v_mul_f32'), 'add', cnt(r'v_add_f32|v_sub_f32'), 'mov', cnt(r'v_mov_b32|v_pk_mov'), 'med3', cnt('v_med3'),
          'maxmin', cnt(r'v_max_f32|v_min_f32'), 'swap', cnt('permlane32_swap'), 'total_v', cnt(r'\n\tv_'))
