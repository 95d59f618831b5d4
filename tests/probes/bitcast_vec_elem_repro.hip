// hipcc 7.2 (clang 22, gfx950): __builtin_bit_cast applied to an ELEMENT of an ext-vector reads element 0.
//   hipcc --offload-arch=gfx950 -O3 --cuda-device-only -S -o - tests/probes/bitcast_vec_elem_repro.hip | grep -A12 '^_Z3bad'
// `bad` compiles to loads of b[0] and b[2] only and `v_add_f32 v, v, v`: the y half is a copy of the x half.  `good` (the elements
// copied to float temporaries first) is the intended two v_pk_add_f32 / two v_max3_i32 / v_pk_mul_f32 / v_add_f32.
// Met in two_opt_common.h box_lb (round 3); the parity tests would have caught the wrong tile bound, the .s did first.
#include <hip/hip_runtime.h>
typedef float v2f __attribute__((ext_vector_type(2)));
__global__ void bad(float px, float py, const float4 *b, float *out)
{
    const float4 box = b[threadIdx.x];
    const v2f p = {px, py}, lo = {box.x, box.y}, hi = {box.z, box.w};
    const v2f u = lo - p, v = p - hi;
    const int ux = __builtin_bit_cast(int, u.x), vx = __builtin_bit_cast(int, v.x);
    const int uy = __builtin_bit_cast(int, u.y), vy = __builtin_bit_cast(int, v.y);  // <- reads u.x / v.x
    const int mx = ux > vx ? ux : vx, my = uy > vy ? uy : vy;
    v2f d = {__builtin_bit_cast(float, mx > 0 ? mx : 0), __builtin_bit_cast(float, my > 0 ? my : 0)};
    d = d * d;
    out[threadIdx.x] = d.x + d.y;
}
__global__ void good(float px, float py, const float4 *b, float *out)
{
    const float4 box = b[threadIdx.x];
    const v2f p = {px, py}, lo = {box.x, box.y}, hi = {box.z, box.w};
    const v2f u = lo - p, v = p - hi;
    const float uxf = u.x, uyf = u.y, vxf = v.x, vyf = v.y;
    const int ux = __builtin_bit_cast(int, uxf), vx = __builtin_bit_cast(int, vxf);
    const int uy = __builtin_bit_cast(int, uyf), vy = __builtin_bit_cast(int, vyf);
    const int mx = ux > vx ? ux : vx, my = uy > vy ? uy : vy;
    v2f d = {__builtin_bit_cast(float, mx > 0 ? mx : 0), __builtin_bit_cast(float, my > 0 ? my : 0)};
    d = d * d;
    out[threadIdx.x] = d.x + d.y;
}
