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
    void (*ks[])(const float*, float*, int) = { k_plain, k_dpp, k_lds, k_readlane, k_lds1 };
    const char* names[] = { "plain v_add chain", "dpp wave_ror chain", "lds broadcast b128", "readlane + add", "lds b128, 1 wait / 64" };
    for (int v = 0; v < 5; ++v) {
        ks[v]<<<1, 64>>>(in, out, 1000);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(a));
        ks[v]<<<1, 64>>>(in, out, iters);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        printf("%-22s %.3f ns/value\n", names[v], ms * 1e6 / (64.0 * iters));
    }
    return 0;
}
