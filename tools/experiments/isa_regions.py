"""Per-region instruction census of one kernel in a hipcc -S listing: regions are delimited by s_barrier.
usage: isa_regions.py <file.s> <kernel-name-substring>"""
import re
import sys

L = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2]
start = [i for i, l in enumerate(L) if pat in l and re.match(r'^_Z\S+:', l)][0]
end = [i for i, l in enumerate(L) if '.amdhsa_kernel' in l and pat in l][0]
F = L[start:end]
bars = [0] + [i for i, l in enumerate(F) if 's_barrier' in l]
cols = [('mfma', 'v_mfma'), ('accrd', 'v_accvgpr_read'), ('accwr', 'v_accvgpr_write'), ('scr', 'scratch_'), ('dsrd', 'ds_read'), ('vpk', r'v_pk_'),
        ('cvt', 'v_cvt'), ('mov', r'v_mov_b32'), ('nop', 's_nop'), ('valu', r'^\s+v_(?!mfma|accvgpr)'), ('salu', r'^\s+s_(?!nop|waitcnt|barrier)'), ('wait', 's_waitcnt')]
print(f"{len(F)} lines; regions between barriers:")
for a, b in zip(bars, bars[1:] + [len(F)]):
    print(f"{a:6d}-{b:6d} " + ' '.join(f"{n}={sum(1 for l in F[a:b] if re.search(p, l))}" for n, p in cols))
