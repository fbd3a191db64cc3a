#!/bin/bash
# device ISA of one source: tools/isa.sh conv_pipe [extra flags] -> /tmp/isa/<name>.s  (resource usage remarks -> /tmp/isa/<name>.remarks)
set -e
cd "$(dirname "$0")/../diffusion-models-made-easy_amd/csrc"
n=$1; shift
mkdir -p /tmp/isa
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -DDMME_BUILD -fvisibility=hidden --cuda-device-only -S "$@" \
    -Rpass-analysis=kernel-resource-usage $n.hip -o /tmp/isa/$n.s 2> /tmp/isa/$n.remarks
echo /tmp/isa/$n.s
