"""Per-basic-block VALU issue-slot estimate of a kernel in a .s file (weights from profiles/r01_microbench.txt:
full-rate f32/int ops 1.0, compare/select/min/max/convert and 3-operand integer 1.7, transcendental 3.3).
usage: python tools/isa_blocks.py file.s mangled_kernel_substring [min_valu]"""
import collections
import re
import sys

FULL = {'v_add_f32', 'v_sub_f32', 'v_mul_f32', 'v_fma_f32', 'v_add_u32', 'v_sub_u32', 'v_and_b32', 'v_or_b32', 'v_xor_b32',
        'v_lshlrev_b32', 'v_lshrrev_b32', 'v_mov_b32', 'v_subrev_u32', 'v_subrev_f32', 'v_ashrrev_i32', 'v_not_b32'}
TRANS = {'v_sqrt_f32', 'v_rcp_f32', 'v_rsq_f32', 'v_rcp_iflag_f32', 'v_exp_f32', 'v_log_f32'}


def base(op):
    return re.sub(r'_(e32|e64|dpp|sdwa)$', '', op)


def cost(c):
    t = 0.0
    for op, n in c.items():
        if op.startswith('v_'):
            b = base(op)
            t += n * (3.3 if b in TRANS else 1.0 if b in FULL else 1.7)
    return t


def main():
    s = open(sys.argv[1]).read()
    key = sys.argv[2]
    minv = int(sys.argv[3]) if len(sys.argv) > 3 else 15
    lines = s.split('\n')
    start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\w*' + re.escape(key) + r'\w*:', l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
    blocks, cur = [], ['entry', collections.Counter()]
    for l in lines[start + 1:end]:
        l = l.strip()
        if re.match(r'^\.LBB\d+_\d+:', l):
            blocks.append(cur)
            cur = [l.split(':')[0], collections.Counter()]
            continue
        if l and not l.startswith((';', '.')):
            cur[1][l.split()[0]] += 1
    blocks.append(cur)
    print(lines[start], end - start, 'lines')
    for name, c in blocks:
        v = sum(n for op, n in c.items() if op.startswith('v_'))
        if v >= minv:
            tr = sum(n for o, n in c.items() if base(o) in TRANS)
            half = sum(n for o, n in c.items() if o.startswith('v_') and base(o) not in TRANS and base(o) not in FULL)
            print(f"{name:10s} valu {v:4d} slots {cost(c):6.1f}  trans {tr:2d} half {half:3d}  ds {sum(n for o, n in c.items() if o.startswith('ds_')):3d} "
                  f"salu {sum(n for o, n in c.items() if o.startswith('s_')):3d} vmem {sum(n for o, n in c.items() if o.startswith(('global_', 'buffer_', 'flat_', 'scratch_'))):2d}")
    tot = collections.Counter()
    for _, c in blocks:
        tot.update(c)
    print('total valu', sum(n for o, n in tot.items() if o.startswith('v_')), 'slots', round(cost(tot), 1))
    print('top half-rate ops:', sorted(((n, o) for o, n in tot.items() if o.startswith('v_') and base(o) not in FULL and base(o) not in TRANS), reverse=True)[:14])


main()
