#!/bin/bash
# Development aid: build libsdmi_hip_<tag>.so with another sdmi_attn.hip configuration (other objects reused as built).
#   tools/ab_attn.sh tagname -DSDMI_ATTN_NST=2 -DSDMI_ATTN_OCC=4 ...      (use with SDMI_LIB=<path>)
set -e
cd "$(dirname "$0")/../stablediffusion_amd/csrc"
tag=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value "$@" -c sdmi_attn.hip -o /tmp/ab_attn_$tag.o
objs=$(ls *.o | grep -v sdmi_attn.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libsdmi_hip_$tag.so $objs /tmp/ab_attn_$tag.o
