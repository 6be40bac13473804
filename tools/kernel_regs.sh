#!/bin/bash
# development: registers / spills / scratch of every kernel in rt_kernels.hip (device-only compile, no GPU needed)
set -e
cd "$(dirname "$0")/.."
mkdir -p /tmp/ks
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Iinclude -Iraytracer_amd/csrc "$@" \
  --cuda-device-only -c raytracer_amd/csrc/rt_kernels.hip -o /tmp/ks/k.co
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=/tmp/ks/k.co --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=/tmp/ks/k.elf
/opt/rocm/lib/llvm/bin/llvm-readobj --notes /tmp/ks/k.elf > /tmp/ks/notes.txt
python3 - <<'PY'
import re
txt = open('/tmp/ks/notes.txt').read()
for blk in re.split(r'\n\s+- \.agpr_count', txt)[1:]:
    g = lambda k: (re.search(r'\.' + k + r':\s+(\S+)', blk) or [None, '?'])[1]
    name = g('name')
    m = re.search(r'(k_[a-z_]+)(I[^E]*E)?', name)
    short = name
    t = re.search(r'\d+(k_[a-z_]+)(?:I((?:Lb[01]E|Li\d+E|\w)*?)E)?v', name)
    if t:
        args = re.findall(r'L[bi](\d+)E', t.group(2) or '')
        short = t.group(1) + ('<' + ','.join(args) + '>' if args else '')
    print(f"{short:28s} vgpr {g('vgpr_count'):>4s} sgpr {g('sgpr_count'):>4s} vspill {g('vgpr_spill_count'):>3s} sspill {g('sgpr_spill_count'):>3s} scratch {g('private_segment_fixed_size'):>4s} lds {g('group_segment_fixed_size'):>5s}")
PY
