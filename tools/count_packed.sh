# Static count of packed FP32 instructions (v_pk_fma_f32, v_pk_mul_f32, v_pk_add_f32) and of the half -> float mixed
# FMAs (v_fma_mix_f32) in the code of the render kernels the workloads run.  No GPU needed.
# usage: bash tools/count_packed.sh
set -e
cd "$(dirname "$0")/../raytracer_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -S --cuda-device-only -o /tmp/rt_count_packed.s rt_kernels.hip 2>/dev/null
for k in "_Z7k_fusedILi0ELb0ELi16ELi8ELb1ELb0ELb0ELi256ELi0E random spheres (cfg2)" \
         "_Z7k_fusedILi1ELb0ELi1ELi5ELb0ELb0ELb1ELi256ELi0E Cornell with smoke, list form (cfg3)" \
         "_Z7k_fusedILi3ELb0ELi24ELi4ELb1ELb0ELb0ELi256ELi0E CSG and textures (cfg4)" \
         "_Z7k_fusedILi0ELb0ELi16ELi6ELb1ELb1ELb0ELi512ELi448E million spheres (cfg5)"; do
  sym=${k%% *}; what=${k#* }
  L=$(grep -n "^$sym" /tmp/rt_count_packed.s | head -1 | cut -d: -f1)
  sed -n "${L},\$p" /tmp/rt_count_packed.s | awk '{print} /s_endpgm/{exit}' > /tmp/rt_count_packed_k.s
  echo "$what: $(grep -c '^\sv_' /tmp/rt_count_packed_k.s) vector ALU instructions, of them $(grep -c 'v_pk_[a-z]*_f32' /tmp/rt_count_packed_k.s) packed FP32 (v_pk_fma/mul/add_f32) and $(grep -c v_fma_mix /tmp/rt_count_packed_k.s) v_fma_mix_f32"
done
