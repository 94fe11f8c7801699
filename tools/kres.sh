#!/bin/bash
# register / LDS / scratch use of the kernels matching a pattern, from the device assembly: tools/kres.sh [pattern] [extra hipcc flags]
P=${1:-k_vote}; shift
D=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only "$@" -o $D/e.s "$(dirname "$0")/../gencore_amd/csrc/engine.hip" 2>/dev/null
python3 - "$D/e.s" "$P" <<'PY'
import re,sys
t=open(sys.argv[1]).read()
md=t[t.index('amdhsa.kernels:'):]
for blk in md.split('  - .agpr_count:')[1:]:
    nm=re.search(r'\.name:\s+(\S+)',blk).group(1)
    if sys.argv[2] in nm:
        g=lambda k: re.search(r'\.%s:\s+(\d+)'%k,blk).group(1)
        print(nm[:50],'vgpr',g('vgpr_count'),'sgpr',g('sgpr_count'),'lds',g('group_segment_fixed_size'),'scratch',g('private_segment_fixed_size'))
PY
cp $D/e.s /tmp/isa/last.s 2>/dev/null; rm -rf $D
