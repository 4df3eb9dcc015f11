#!/bin/bash
# s_waitcnt vmcnt(N) values, s_setprio and branch counts of the kernels whose mangled
# name contains $1 (default: the fp32 product sweep).  A hot loop that waits vmcnt(0)
# everywhere has lost its prefetch window.
pat=${1:-stress_grad_kernelIfLb1ELb1ELi0ELb1ELi}; so=${2:-blueberry_amd/libblueberry_hip.so}
tmp=$(mktemp -d); B=/opt/rocm/lib/llvm/bin
$B/llvm-objcopy -O binary --only-section=.hip_fatbin $so $tmp/fat.bin
$B/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$tmp/fat.bin --output=$tmp/k.co --unbundle
$B/llvm-objdump -d $tmp/k.co > $tmp/k.s
python3 - $tmp/k.s "$pat" <<'PY'
import sys,re
txt=open(sys.argv[1]).read()
for m in re.finditer(r'^[0-9a-f]+ <(\S+)>:\n(.*?)(?=^[0-9a-f]+ <|\Z)', txt, re.S|re.M):
    name=m.group(1)
    if sys.argv[2] not in name: continue
    lines=m.group(2).splitlines()
    waits=[re.search(r'vmcnt\((\d+)\)', l).group(1) for l in lines if 's_waitcnt' in l and 'vmcnt' in l]
    print(name[:60], '...', name[-14:], 'instr', len(lines), 'vmcnt', waits,
          'setprio', sum('s_setprio' in l for l in lines), 'branches', sum('s_cbranch' in l or 's_branch' in l for l in lines),
          'scratch', sum('scratch_' in l for l in lines))
PY
rm -rf $tmp
