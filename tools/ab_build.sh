#!/bin/bash
# Development aid: build a second library with extra -D flags for A/B runs, e.g.
#   tools/ab_build.sh -DSDMI_SETPRIO=0   ->  stablediffusion_amd/csrc/libsdmi_hip_b.so  (use with SDMI_LIB=<path>)
set -e
cd "$(dirname "$0")/../stablediffusion_amd/csrc"
objs=""
for f in sdmi_api sdmi_gemm sdmi_gemm_m0_bf16 sdmi_gemm_m0_f16 sdmi_gemm_m1_bf16 sdmi_gemm_m1_f16 sdmi_gemm_wide_bf16 sdmi_gemm_wide_f16 sdmi_gemm_panel sdmi_conv_narrow sdmi_conv_halo sdmi_norm sdmi_elem sdmi_attn \
         sdmi_attn_vae sdmi_f32 sdmi_pack sdmi_probe sdmi_gemm_lnf sdmi_xattn; do
  rm -f /tmp/ab_$f.o      # (a failed compile must not leave an older object for the link)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value "$@" -c $f.hip -o /tmp/ab_$f.o &
  objs="$objs /tmp/ab_$f.o"
  if (( $(jobs -r | wc -l) >= 6 )); then wait -n || true; fi      # six compiles at a time (8 CPUs, 64 GiB)
done
wait
for o in $objs; do [ -f $o ] || { echo "ab_build: $o was not built" >&2; exit 1; }; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libsdmi_hip_b.so $objs
