#!/bin/bash
# Registers / scratch / LDS of every kernel in the product .so (from the code object's
# metadata), and per-kernel instruction counts of interest.  Runs without a GPU.
so=${1:-blueberry_amd/libblueberry_hip.so}
tmp=$(mktemp -d); B=/opt/rocm/lib/llvm/bin
$B/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$so --output=$tmp/k.co --unbundle 2>/dev/null \
  || /opt/rocm/bin/roc-obj-ls $so >/dev/null 2>&1
if [ ! -s $tmp/k.co ]; then
  # the .so embeds the fat binary in .hip_fatbin: pull it out and unbundle that
  $B/llvm-objcopy -O binary --only-section=.hip_fatbin $so $tmp/fat.bin
  $B/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$tmp/fat.bin --output=$tmp/k.co --unbundle
fi
$B/llvm-readelf --notes $tmp/k.co | python3 -c "
import sys,re
txt=sys.stdin.read()
for m in re.finditer(r'\.name:\s+(\S+).*?(?=\.name:|\Z)', txt, re.S):
    blk=m.group(0)
    if '.vgpr_count' not in blk: continue
    g=lambda k: (re.search(r'\.'+k+r':\s+(\d+)', blk) or [0,'?'])[1]
    print('%-4s vgpr %-4s sgpr %-5s scratch %-6s lds  %s' % (g('vgpr_count'), g('sgpr_count'), g('private_segment_fixed_size'), g('group_segment_fixed_size'), m.group(1)[:110]))
" | sort -k9 | c++filt | cut -c1-220
if [ -n "$2" ]; then
  $B/llvm-objdump -d $tmp/k.co > $tmp/k.s
  python3 - "$tmp/k.s" "$2" <<'PY'
import sys,re,collections
txt=open(sys.argv[1]).read(); pat=sys.argv[2]
for m in re.finditer(r'^[0-9a-f]+ <(\S+)>:\n(.*?)(?=^[0-9a-f]+ <|\Z)', txt, re.S|re.M):
    if pat not in m.group(1): continue
    c=collections.Counter()
    for line in m.group(2).splitlines():
        t=line.split()
        if t: c[t[0]]+=1
    keys=['v_pk_add_f32','v_pk_mul_f32','v_pk_fma_f32','v_rsq_f32','v_rsq_f64','v_fma_f64','v_mul_f64','v_add_f64','v_add_f32_dpp','v_mov_b32_dpp','global_load_dwordx4','buffer_store_dword','ds_write_b32','s_waitcnt','scratch_load_dword','scratch_store_dword','s_load_dwordx8','v_readlane_b32','s_nop']
    print(m.group(1)[:100], 'total', sum(c.values()), {k:c[k] for k in keys if c[k]})
PY
fi
rm -rf $tmp
