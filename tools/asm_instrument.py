#!/usr/bin/env python3
"""tools/asm_instrument.py IN.s OUT.s MODE --dump IDX:REG[,REG..] ...   (MODE = dump | cal)
Register dumps for tools/dump_diag.py (round 5, profiles/r05_wrong_value_traced.txt).  Instruments the no-recording / no-auto-reset ActorBF16Lean rollout kernel of the
listing IN.s (variant F of tools/hazard_variants.sh: hipcc -S --cuda-device-only of rq_kernels_16bit.hip; the instruction indices are those
of `tools/asm_edit.py --list`): at instruction index IDX (before it) store the listed VGPRs to scratch slots (512 + 4 k, k counting
over all dumps in argument order); in the epilogue, each of the 20 state-column stores gets its data register reloaded from slot k first
(MODE dump) or set to the constant k + 1 (MODE cal: which output column is which slot)."""
import re, struct, sys
K = 'k_rollout_fusedILb0ELb0ELb0ELb0ENS_13ActorBF16Lean'
src, out, mode = sys.argv[1], sys.argv[2], sys.argv[3]
dumps = []
args = sys.argv[4:]
while args:
    assert args[0] == '--dump'
    idx, regs = args[1].split(':')
    dumps.append((int(idx), [int(r) for r in regs.split(',')]))
    args = args[2:]
lines = open(src).read().split('\n')
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l) and K in l)
name = lines[start].split(':')[0]
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('s_endpgm'))
def is_ins(l):
    s = l.strip()
    return l.startswith('\t') and s and not s.startswith(('.', ';', '//'))
index = {}
n = 0
for i in range(start, end + 1):
    if is_ins(lines[i]):
        index[n] = i; n += 1
STORE_REGS = [116, 117, 229, 112, 114, 115, 113, 110, 111, 108, 106, 107, 109, 104, 105, 102, 103, 181, 118, 119]
ins_before = {}            # line -> [text]
slot = 0
for idx, regs in dumps:
    t = []
    for r in regs:
        t.append(f'\tscratch_store_dword off, v{r}, off offset:{512 + 4 * slot}')
        slot += 1
    ins_before.setdefault(index[idx], []).extend(t)
assert slot <= 20
# epilogue
k = 0
for i in range(index[2420], end):
    m = re.match(r'\tglobal_store_dword v\[\d+:\d+\], v(\d+), off\s*$', lines[i])
    if m and k < 20 and int(m.group(1)) == STORE_REGS[k]:
        r = STORE_REGS[k]
        if mode == 'cal':
            bits = struct.unpack('<I', struct.pack('<f', float(k + 1)))[0]
            ins_before.setdefault(i, []).append(f'\tv_mov_b32 v{r}, 0x{bits:08x}')
            ins_before[i].append('\ts_nop 1')
        elif k < slot:
            ins_before.setdefault(i, []).extend([f'\tscratch_load_dword v{r}, off, off offset:{512 + 4 * k}', '\ts_waitcnt vmcnt(0)'])
        k += 1
assert k == 20, k
res = []
for i, l in enumerate(lines):
    if i in ins_before:
        res.extend(ins_before[i])
    res.append(l)
text = '\n'.join(res)
# scratch size: kernel descriptor and metadata
a = text.index('.amdhsa_kernel ' + name)
b = text.index('.end_amdhsa_kernel', a)
blk = text[a:b]
blk2 = re.sub(r'\.amdhsa_private_segment_fixed_size \d+', '.amdhsa_private_segment_fixed_size 1024', blk)
assert blk != blk2
text = text[:a] + blk2 + text[b:]
m = text.index('.name:           ' + name + '\n')
# the metadata entry: fields around .name; find the enclosing '  - .agpr_count' block
s0 = text.rfind('  - .agpr_count', 0, m)
s1 = text.find('  - .agpr_count', m)
if s1 < 0: s1 = text.find('amdhsa.target', m)
ent = text[s0:s1]
ent2 = re.sub(r'\.private_segment_fixed_size: \d+', '.private_segment_fixed_size: 1024', ent)
assert ent != ent2
text = text[:s0] + ent2 + text[s1:]
open(out, 'w').write(text)
print(out, 'slots', slot, 'mode', mode, file=sys.stderr)
