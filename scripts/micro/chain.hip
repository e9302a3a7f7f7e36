// Microbenchmark: cost per value of a sequential fp32 add chain in ONE wave (k_split_value's floor).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_plain(const float* in, float* out, int iters)
{
    float x = in[threadIdx.x], acc = 0.0f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 64; ++j) acc += x;
    }
    out[threadIdx.x] = acc;
}

// acc travels one lane per step: new[l] = acc[l-1] + x[l]
__global__ void k_dpp(const float* in, float* out, int iters)
{
    float x = in[threadIdx.x], acc = 0.0f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            const float prev = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc), 0x13C /* wave_ror:1 */, 0xF, 0xF, false));
            acc = prev + x;
        }
    }
    out[threadIdx.x] = acc;
}

__global__ void k_lds(const float* in, float* out, int iters)
{
    __shared__ __attribute__((aligned(16))) float buf[256];
    for (int j = 0; j < 4; ++j) buf[threadIdx.x + 64 * j] = in[threadIdx.x];
    __syncthreads();
    float acc = 0.0f;
    const float4* q4 = reinterpret_cast<const float4*>(buf);
    for (int i = 0; i < iters; ++i) {
        float4 t[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) t[q] = q4[(i & 3) * 16 + q];
#pragma unroll
        for (int q = 0; q < 16; ++q) { acc += t[q].x; acc += t[q].y; acc += t[q].z; acc += t[q].w; }
    }
    out[threadIdx.x] = acc;
}

// same, with one explicit wait for all sixteen reads instead of one per read
__global__ void k_lds1(const float* in, float* out, int iters)
{
    __shared__ __attribute__((aligned(16))) float buf[256];
    for (int j = 0; j < 4; ++j) buf[threadIdx.x + 64 * j] = in[threadIdx.x];
    __syncthreads();
    float acc = 0.0f;
    const float4* q4 = reinterpret_cast<const float4*>(buf);
    float4 t[16], u[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) t[q] = q4[q];
    for (int i = 0; i < iters; i += 2) {
#pragma unroll
        for (int q = 0; q < 16; ++q) u[q] = q4[((i + 1) & 3) * 16 + q];
#pragma unroll
        for (int q = 0; q < 16; ++q) { acc += t[q].x; acc += t[q].y; acc += t[q].z; acc += t[q].w; }
        __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll
        for (int q = 0; q < 16; ++q) t[q] = q4[((i + 2) & 3) * 16 + q];
#pragma unroll
        for (int q = 0; q < 16; ++q) { acc += u[q].x; acc += u[q].y; acc += u[q].z; acc += u[q].w; }
        __builtin_amdgcn_s_waitcnt(0xC07F);
    }
    out[threadIdx.x] = acc;
}

// values stay in the registers the coalesced load put them in (lane l holds value l): lane 15 of each
// 16-lane row adds its row's values in order through DPP row_shr on the VALUE operand (no hazard: the
// value register is old), row_bcast:15 hands the sum to the next row, readlane 63 to the next register
template <int N> __device__ __forceinline__ float shr(float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x110 + N, 0xF, 0xF, true));
}
__device__ __forceinline__ float row_sum(float acc, float x)
{
    acc += shr<15>(x); acc += shr<14>(x); acc += shr<13>(x); acc += shr<12>(x);
    acc += shr<11>(x); acc += shr<10>(x); acc += shr<9>(x);  acc += shr<8>(x);
    acc += shr<7>(x);  acc += shr<6>(x);  acc += shr<5>(x);  acc += shr<4>(x);
    acc += shr<3>(x);  acc += shr<2>(x);  acc += shr<1>(x);  acc += x;
    return acc;
}
template <int ROWMASK> __device__ __forceinline__ float next_row(float acc)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, acc), __builtin_bit_cast(int, acc), 0x142, ROWMASK, 0xF, false));
}
__device__ __forceinline__ float reg_sum(float acc, float x)
{
    acc = row_sum(acc, x);            // lane 15
    acc = next_row<0x2>(acc);
    acc = row_sum(acc, x);            // lane 31
    acc = next_row<0x4>(acc);
    acc = row_sum(acc, x);            // lane 47
    acc = next_row<0x8>(acc);
    acc = row_sum(acc, x);            // lane 63
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, acc), 63));
}
__global__ void k_dpp_rows(const float* in, float* out, int iters)
{
    float x = in[threadIdx.x], acc = 0.0f;
    for (int i = 0; i < iters; ++i) acc = reg_sum(acc, x);
    out[threadIdx.x] = acc;
}

__global__ void k_readlane(const float* in, float* out, int iters)
{
    float x = in[threadIdx.x], acc = 0.0f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 64; ++j) acc += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), j));
    }
    out[threadIdx.x] = acc;
}

int main()
{
    float *in, *out;
    CHECK(hipMalloc(&in, 1024)); CHECK(hipMalloc(&out, 1024));
    CHECK(hipMemset(in, 0, 1024));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const int iters = 200000; // x 64 values
    void (*ks[])(const float*, float*, int) = { k_plain, k_dpp, k_lds, k_readlane, k_lds1, k_dpp_rows };
    const char* names[] = { "plain v_add chain", "dpp wave_ror chain", "lds broadcast b128", "readlane + add", "lds b128, 1 wait / 64", "dpp row_shr on values" };
    for (int v = 0; v < 6; ++v) {
        ks[v]<<<1, 64>>>(in, out, 1000);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(a));
        ks[v]<<<1, 64>>>(in, out, iters);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        printf("%-22s %.3f ns/value\n", names[v], ms * 1e6 / (64.0 * iters));
    }
    // the row chain must equal the sequential sum bit for bit
    float h[64], ref = 0.0f;
    for (int i = 0; i < 64; ++i) h[i] = 1.0f / (float)(3 * i + 1) * ((i % 3) ? 1e3f : -1e-3f);
    CHECK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
    for (int r = 0; r < 5; ++r) for (int i = 0; i < 64; ++i) ref += h[i];
    k_dpp_rows<<<1, 64>>>(in, out, 5);
    float got[64];
    CHECK(hipMemcpy(got, out, sizeof(got), hipMemcpyDeviceToHost));
    printf("dpp rows check: got %a ref %a %s\n", got[0], ref, got[0] == ref ? "OK" : "MISMATCH");
    return 0;
}
