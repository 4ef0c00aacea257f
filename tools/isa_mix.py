#!/usr/bin/env python3
"""Instruction mix of every kernel in a `hipcc -S` listing: total, MFMA, packed-fp32, transcendental, conversions, selects,
24-bit multiplies, LDS and memory instructions, scratch use.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast --cuda-device-only -S csrc/kernels/attention.hip -o /tmp/attn.s
    python tools/isa_mix.py /tmp/attn.s [substring of the kernel name]
"""
import collections
import re
import sys


def main() -> None:
    txt = open(sys.argv[1]).read()
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    heads = list(re.finditer(r"^(_Z\w+):", txt, re.M))
    print(f"{'instr':>6} {'mfma':>5} {'v_pk':>5} {'exp':>4} {'cvt':>4} {'cnd':>4} {'mul24':>5} {'ds':>4} {'vmem':>5} {'valu':>5} {'nop':>4}  kernel")
    for i, m in enumerate(heads):
        name = m.group(1)
        if want not in name:
            continue
        body = txt[m.end():heads[i + 1].start() if i + 1 < len(heads) else len(txt)].split(".Lfunc_end")[0]
        ops = collections.Counter(l.split()[0] for l in body.split("\n")
                                  if l.startswith("\t") and l.split() and not l.startswith(("\t.", "\t;")))
        def n(pred):
            return sum(v for k, v in ops.items() if pred(k))
        print(f"{sum(ops.values()):6d} {n(lambda k: 'mfma' in k):5d} {n(lambda k: k.startswith('v_pk_')):5d} "
              f"{n(lambda k: k.startswith('v_exp')):4d} {n(lambda k: 'cvt' in k):4d} {n(lambda k: 'cndmask' in k):4d} "
              f"{n(lambda k: 'u32_u24' in k):5d} {n(lambda k: k.startswith('ds_')):4d} "
              f"{n(lambda k: k.startswith(('global_', 'buffer_', 'scratch_'))):5d} {n(lambda k: k.startswith('v_') and 'mfma' not in k):5d} {n(lambda k: k == 's_nop'):4d}  {name[:90]}")


if __name__ == "__main__":
    main()
