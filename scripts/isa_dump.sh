#!/bin/bash
# Disassembly of one kernel of ptk.hip for the current sources, with an instruction mix and the loops found in it (backward
# branches): what the instruction-level work in DESIGN.md §4 was read from.
# Usage: scripts/isa_dump.sh <substring of the mangled kernel name, e.g. k_walkILb0 or k_bounce_fusedILi40> [out.s] [extra hipcc flags...]
set -e
cd "$(dirname "$0")/.."
key=$1; out=${2:-/tmp/ptk_kernel.s}; shift; [ $# -gt 0 ] && shift
flags=$(python3 -c "import __graft_entry__ as g; print(' '.join(f for f in g.HIPCC_FLAGS if f not in ('-shared', '-fPIC')))")
tmp=$(mktemp -d)
/opt/rocm/bin/hipcc $flags --cuda-device-only -c "$@" -o $tmp/ptk.co pathtracing_amd/csrc/ptk.hip
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$tmp/ptk.co --output=$tmp/ptk.elf --unbundle
/opt/rocm/lib/llvm/bin/llvm-objdump -d $tmp/ptk.elf > $tmp/ptk.s
python3 - "$tmp/ptk.s" "$key" "$out" <<'P'
import collections, re, sys
cur = None
lines = collections.defaultdict(list)
for line in open(sys.argv[1]):
    m = re.match(r'^[0-9a-f]+ <(.*)>:', line)
    if m:
        cur = m.group(1)
        continue
    if cur and '\t' in line:
        lines[cur].append(re.sub(r'\s*//.*$', '', line).strip())
names = [n for n in lines if sys.argv[2] in n]
if not names:
    sys.exit("no kernel matches " + sys.argv[2])
L = lines[names[0]]
open(sys.argv[3], 'w').write('\n'.join(f'{i} {l}' for i, l in enumerate(L)) + '\n')
c = collections.Counter(l.split()[0] for l in L)
grp = lambda p: sum(n for i, n in c.items() if i.startswith(p))
print(names[0])
print(f'{len(L)} instructions: VALU {grp("v_")} (v_mov {c["v_mov_b32_e32"]}), SALU {grp("s_")}, LDS {grp("ds_")}, global {grp("global_")}, scratch {grp("scratch_")}')
print('quieting v_max x, x:', sum(1 for l in L if re.match(r'v_max_f32_e32 v\d+, (v\d+), \1$', l)),
      ' conditions materialised as 0 / 1:', sum(1 for l in L if re.match(r'v_cndmask_b32_e64 v\d+, 0, 1, ', l)))
for i, l in enumerate(L):
    m = re.match(r's_cbranch_\w+ (\d+)|s_branch (\d+)', l)
    if m and int(m.group(1) or m.group(2)) > 32767:
        t = i + 1 - (65536 - int(m.group(1) or m.group(2)))
        if t < 0:
            continue
        b = L[t:i + 1]
        print(f'loop {t}..{i}: {len(b)} instructions, VALU {sum(1 for x in b if x.startswith("v_"))} (v_mov {sum(1 for x in b if x.startswith("v_mov"))}), '
              f'loads {sum(1 for x in b if x.startswith(("global_load", "s_load")))}, LDS {sum(1 for x in b if x.startswith("ds_"))}')
print('written to', sys.argv[3])
P
rm -rf $tmp
