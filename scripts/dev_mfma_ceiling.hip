// Dev experiment (not part of the library): what a 112 x 64 wave tile of v_mfma_f32_16x16x32_bf16 sustains on random bf16
// operands on THIS device, (0) with every operand in registers, (1) with the patch kernels' 11 ds_read_b128 per 28 MFMAs,
// (2) like (1) with 2 waves of the 4 per SIMD pair only reading (the LOAD / COMPUTE split).  The quotient of the conv
// kernels' rate and these is how far they are from what the chip's clock control gives a loop of this shape.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_ceiling scripts/dev_mfma_ceiling.hip && /tmp/mfma_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(512) void loop_kernel(const bf16x8* __restrict__ src, float* __restrict__ out, int iters,
                                                   unsigned long long* __restrict__ stamps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    bf16x8* lds = reinterpret_cast<bf16x8*>(smem);
    // 64 KB of random operands in LDS (4096 x 16 B)
    for (int i = tid; i < 4096; i += 512) lds[i] = src[(blockIdx.x & 15) * 4096 + i];
    __syncthreads();
    bf16x8 a[7], b[4];
    for (int i = 0; i < 7; ++i) a[i] = lds[(tid * 7 + i) & 4095];
    for (int i = 0; i < 4; ++i) b[i] = lds[(tid * 4 + i + 1777) & 4095];
    f32x4 acc[28];
    for (int i = 0; i < 28; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    unsigned off = lane;   // conflict-free: 64 lanes x 16 B consecutive
    for (int it = 0; it < iters; ++it) {
        if (MODE >= 1) {
            for (int i = 0; i < 7; ++i) a[i] = lds[(off + 64 * i) & 4095];
            for (int i = 0; i < 4; ++i) b[i] = lds[(off + 64 * (i + 7)) & 4095];
            off += 64 * 11;
        } else {
            asm volatile("" : "+v"(a[0]), "+v"(b[0]));
        }
        for (int i = 0; i < 7; ++i)
            for (int j = 0; j < 4; ++j)
                acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i * 4 + j], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    f32x4 s = acc[0];
    for (int i = 1; i < 28; ++i) s += acc[i];
    out[blockIdx.x * 512 + tid] = s[0] + s[1] + s[2] + s[3];
    if (tid == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

// software-pipelined: the next step's NA + NB fragments are read while this step's NA x NB MFMAs issue
template <int NA, int NB, int THREADS, int PER = 0>
__global__ __launch_bounds__(THREADS) void pipe_kernel(const bf16x8* __restrict__ src, float* __restrict__ out, int iters,
                                                       unsigned long long* __restrict__ stamps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    bf16x8* lds = reinterpret_cast<bf16x8*>(smem);
    for (int i = tid; i < 4096; i += THREADS) lds[i] = src[(blockIdx.x & 15) * 4096 + i];
    __syncthreads();
    bf16x8 a[2][NA], b[2][NB];
    unsigned off = lane + 64 * (tid >> 6);
    for (int i = 0; i < NA; ++i) a[0][i] = lds[(off + 64 * i) & 4095];
    for (int i = 0; i < NB; ++i) b[0][i] = lds[(off + 64 * (i + NA)) & 4095];
    f32x4 acc[NA * NB];
    for (int i = 0; i < NA * NB; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            off += 64 * (NA + NB);
#pragma unroll
            for (int i = 0; i < NA; ++i) a[h ^ 1][i] = lds[(off + 64 * i) & 4095];
#pragma unroll
            for (int i = 0; i < NB; ++i) b[h ^ 1][i] = lds[(off + 64 * (i + NA)) & 4095];
#pragma unroll
            for (int i = 0; i < NA; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    acc[i * NB + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[h][i], b[h][j], acc[i * NB + j], 0, 0, 0);
            if (PER > 0) {   // one LDS read, then PER MFMAs, ...
#pragma unroll
                for (int g = 0; g < NA + NB; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
                }
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    f32x4 s = acc[0];
    for (int i = 1; i < NA * NB; ++i) s += acc[i];
    out[blockIdx.x * THREADS + tid] = s[0] + s[1] + s[2] + s[3];
    if (tid == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

// row-wise: A fragment i is re-read for the next step right after its row of NB MFMAs, the B fragments are double-buffered
// and read one or two per row; addresses are one base register + immediates
// BAR: 0 no barrier; 1 one s_barrier at the end of every step; 2 two per step (after row NA/2 and at the end), the second
// wave group started half a step late; 3 one every second step; 4 like 1 with the second group half a step late
template <int NA, int NB, int THREADS, int BAR = 0>
__global__ __launch_bounds__(THREADS) void row_kernel(const bf16x8* __restrict__ src, float* __restrict__ out, int iters,
                                                      unsigned long long* __restrict__ stamps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    bf16x8* lds = reinterpret_cast<bf16x8*>(smem);
    for (int i = tid; i < 4096; i += THREADS) lds[i] = src[(blockIdx.x & 15) * 4096 + i];
    __syncthreads();
    bf16x8 a[NA], b[2][NB];
    const bf16x8* p0 = lds + lane;
    for (int i = 0; i < NA; ++i) a[i] = p0[64 * i];
    for (int i = 0; i < NB; ++i) b[0][i] = p0[64 * (i + NA)];
    f32x4 acc[NA * NB];
    for (int i = 0; i < NA * NB; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    constexpr int BPR = (NB + NA - 1) / NA;   // B reads per row
    const bool late = (BAR == 2 || BAR == 4) && __builtin_amdgcn_readfirstlane(tid >> 6) >= THREADS / 128;
    if (BAR == 2 && late) __builtin_amdgcn_s_barrier();
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const bf16x8* p = p0 + (h ? 0 : 2048 - 64 * (NA + NB));
#pragma unroll
            for (int i = 0; i < NA; ++i) {
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    acc[i * NB + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[h][j], acc[i * NB + j], 0, 0, 0);
                a[i] = p[64 * i];
#pragma unroll
                for (int q = 0; q < BPR; ++q)
                    if (i * BPR + q < NB) b[h ^ 1][i * BPR + q] = p[64 * (NA + i * BPR + q)];
                __builtin_amdgcn_sched_barrier(0);
                if (BAR == 2 && i == NA / 2) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
                if (BAR == 4 && i == NA / 2 && late) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
            }
            if (BAR == 1 || BAR == 2 || (BAR == 3 && h == 1) || (BAR == 4 && !late)) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
        }
    }
    if (BAR == 2 && !late) __builtin_amdgcn_s_barrier();
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    f32x4 s = acc[0];
    for (int i = 1; i < NA * NB; ++i) s += acc[i];
    out[blockIdx.x * THREADS + tid] = s[0] + s[1] + s[2] + s[3];
    if (tid == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int NA, int NB, int THREADS, int PER = 0>
static void run_pipe(const char* name, const bf16x8* src, float* out, unsigned long long* stamps, int nblk) {
    const int iters = 20000 * 28 / (NA * NB);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto k = PER < 0 ? &row_kernel<NA, NB, THREADS, (PER < 0 ? -1 - PER : 0)> : &pipe_kernel<NA, NB, THREADS, (PER < 0 ? 0 : PER)>;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    float ms = 0;
    do {
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) k<<<nblk, THREADS, 65536>>>(src, out, iters, stamps);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float m; CHECK(hipEventElapsedTime(&m, e0, e1)); ms += m;
    } while (ms < 2500.f);
    const int reps = 20;
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) k<<<nblk, THREADS, 65536>>>(src, out, iters, stamps);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> st(2 * nblk);
    CHECK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> clk(nblk);
    for (int i = 0; i < nblk; ++i) clk[i] = double(st[2 * i]) / double(st[2 * i + 1]) * 0.1;
    std::sort(clk.begin(), clk.end());
    double flop = double(reps) * nblk * (THREADS / 64) * double(iters) * NA * NB * 16 * 16 * 32 * 2;
    double cyc = double(st[0]) / (double(iters) * NA * NB);
    printf("%-44s %8.1f TFLOP/s  in-kernel clock %.3f GHz  %.2f cycles per MFMA and wave  %.2f ms/launch\n", name,
           flop / (ms * 1e-3) * 1e-12, clk[nblk / 2], cyc, ms / reps);
    fflush(stdout);
}


// ---- v_mfma_f32_32x32x16_bf16: the same 4-register operands, twice the FLOPs per instruction (16 passes), i.e. half the
// operand-register reads and instruction issues per FLOP.  ROW = 0: registers only; 1: row-wise LDS reads as row_kernel
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int NA, int NB, int THREADS, int ROW>
__global__ __launch_bounds__(THREADS) void k32_kernel(const bf16x8* __restrict__ src, float* __restrict__ out, int iters,
                                                      unsigned long long* __restrict__ stamps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    bf16x8* lds = reinterpret_cast<bf16x8*>(smem);
    for (int i = tid; i < 4096; i += THREADS) lds[i] = src[(blockIdx.x & 15) * 4096 + i];
    __syncthreads();
    bf16x8 a[NA], b[2][NB];
    const bf16x8* p0 = lds + lane;
    for (int i = 0; i < NA; ++i) a[i] = p0[64 * i];
    for (int i = 0; i < NB; ++i) b[0][i] = b[1][i] = p0[64 * (i + NA)];
    f32x16 acc[NA * NB];
    for (int i = 0; i < NA * NB; ++i) for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    constexpr int BPR = (NB + NA - 1) / NA;
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const bf16x8* p = p0 + (h ? 0 : 2048 - 64 * (NA + NB));
#pragma unroll
            for (int i = 0; i < NA; ++i) {
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    acc[i * NB + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[h][j], acc[i * NB + j], 0, 0, 0);
                if (ROW) {
                    a[i] = p[64 * i];
#pragma unroll
                    for (int q = 0; q < BPR; ++q)
                        if (i * BPR + q < NB) b[h ^ 1][i * BPR + q] = p[64 * (NA + i * BPR + q)];
                } else {
                    asm volatile("" : "+v"(a[i]));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < NA * NB; ++i) for (int q = 0; q < 16; ++q) s += acc[i][q];
    out[blockIdx.x * THREADS + tid] = s;
    if (tid == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int NA, int NB, int THREADS, int ROW>
static void run32(const char* name, const bf16x8* src, float* out, unsigned long long* stamps, int nblk) {
    const int iters = 20000 * 14 / (NA * NB);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto k = &k32_kernel<NA, NB, THREADS, ROW>;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    float ms = 0;
    do {
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) k<<<nblk, THREADS, 65536>>>(src, out, iters, stamps);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float m; CHECK(hipEventElapsedTime(&m, e0, e1)); ms += m;
    } while (ms < 2500.f);
    const int reps = 20;
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) k<<<nblk, THREADS, 65536>>>(src, out, iters, stamps);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> st(2 * nblk);
    CHECK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> clk(nblk);
    for (int i = 0; i < nblk; ++i) clk[i] = double(st[2 * i]) / double(st[2 * i + 1]) * 0.1;
    std::sort(clk.begin(), clk.end());
    double flop = double(reps) * nblk * (THREADS / 64) * double(iters) * NA * NB * 32 * 32 * 16 * 2;
    double cyc = double(st[0]) / (double(iters) * NA * NB);
    printf("%-44s %8.1f TFLOP/s  in-kernel clock %.3f GHz  %.2f cycles per MFMA and wave  %.2f ms/launch\n", name,
           flop / (ms * 1e-3) * 1e-12, clk[nblk / 2], cyc, ms / reps);
    fflush(stdout);
}

template <int MODE>
static void run(const char* name, const bf16x8* src, float* out, unsigned long long* stamps, int nblk) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&loop_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    // >= 2 s of back-to-back launches before the timed ones
    int warm = 0;
    float ms = 0;
    do {
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) loop_kernel<MODE><<<nblk, 512, 65536>>>(src, out, iters, stamps);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float m; CHECK(hipEventElapsedTime(&m, e0, e1)); ms += m; ++warm;
    } while (ms < 2500.f);
    const int reps = 20;
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) loop_kernel<MODE><<<nblk, 512, 65536>>>(src, out, iters, stamps);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> st(2 * nblk);
    CHECK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> clk(nblk);
    for (int i = 0; i < nblk; ++i) clk[i] = double(st[2 * i]) / double(st[2 * i + 1]) * 0.1;   // GHz (100 MHz real-time)
    std::sort(clk.begin(), clk.end());
    double flop = double(reps) * nblk * 8.0 * iters * 28.0 * 16 * 16 * 32 * 2;
    printf("%-44s %8.1f TFLOP/s  in-kernel clock %.3f GHz (median of %d blocks)  %.2f ms/launch\n", name,
           flop / (ms * 1e-3) * 1e-12, clk[nblk / 2], nblk, ms / reps);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int nblk = 256;
    const bool only32 = argc > 1;
    std::vector<unsigned short> h(16 * 4096 * 8);
    srand(1);
    for (auto& v : h) {
        // random bf16 in (-1, 1): sign, exponent 119..126, 7 mantissa bits
        unsigned s = rand() & 1, e = 119 + (rand() & 7), m = rand() & 127;
        v = (unsigned short)((s << 15) | (e << 7) | m);
    }
    bf16x8* src; float* out; unsigned long long* stamps;
    CHECK(hipMalloc(&src, h.size() * 2)); CHECK(hipMalloc(&out, nblk * 512 * 4)); CHECK(hipMalloc(&stamps, nblk * 16));
    CHECK(hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    run<0>("registers only, 2 waves/SIMD", src, out, stamps, nblk);
    run32<4, 2, 512, 0>("32x32x16 registers only, 4x2, 2 waves/SIMD", src, out, stamps, nblk);
    run32<3, 2, 512, 0>("32x32x16 registers only, 3x2, 2 waves/SIMD", src, out, stamps, nblk);
    run32<4, 2, 512, 1>("32x32x16 row-wise 4x2 (6 per 8), 2 waves/SIMD", src, out, stamps, nblk);
    run32<3, 2, 512, 1>("32x32x16 row-wise 3x2 (5 per 6), 2 waves/SIMD", src, out, stamps, nblk);
    run32<2, 2, 512, 1>("32x32x16 row-wise 2x2 (4 per 4), 2 waves/SIMD", src, out, stamps, nblk);
    run_pipe<7, 4, 512, -1>("row-wise, 7x4, 2 waves/SIMD", src, out, stamps, nblk);
    run_pipe<4, 4, 512, -1>("row-wise, 4x4, 2 waves/SIMD", src, out, stamps, nblk);
    run32<4, 2, 512, 1>("32x32x16 row-wise 4x2 again", src, out, stamps, nblk);
    run<0>("registers only, 2 waves/SIMD again", src, out, stamps, nblk);
    if (only32) return 0;
    run<1>("11 ds_read_b128 per 28 MFMAs, 2 waves/SIMD", src, out, stamps, nblk);
    run<0>("registers only (again)", src, out, stamps, nblk);
    run_pipe<7, 4, 512>("pipelined reads, 7x4 (11 per 28), 2 waves/SIMD", src, out, stamps, nblk);
    run_pipe<4, 4, 512>("pipelined reads, 4x4 (8 per 16), 2 waves/SIMD", src, out, stamps, nblk);
    run_pipe<7, 4, 512, 3>("interleaved 1:3, 7x4, 2 waves/SIMD", src, out, stamps, nblk);
    run_pipe<7, 6, 512, 4>("interleaved 1:4, 7x6, 2 waves/SIMD", src, out, stamps, nblk);
    run_pipe<7, 4, 512, -1>("row-wise, 7x4, 2 waves/SIMD", src, out, stamps, nblk);
    run_pipe<7, 4, 512, -2>("row-wise, 7x4, barrier per step", src, out, stamps, nblk);
    run_pipe<7, 4, 512, -3>("row-wise, 7x4, 2 barriers per step, staggered groups", src, out, stamps, nblk);
    run_pipe<7, 4, 512, -4>("row-wise, 7x4, barrier every 2nd step", src, out, stamps, nblk);
    run_pipe<7, 4, 512, -5>("row-wise, 7x4, barrier per step, staggered groups", src, out, stamps, nblk);
    run_pipe<7, 4, 512, -1>("row-wise, 7x4, 2 waves/SIMD again", src, out, stamps, nblk);
    run_pipe<7, 6, 512, -1>("row-wise, 7x6, 2 waves/SIMD", src, out, stamps, nblk);
    run_pipe<7, 4, 512>("pipelined reads, 7x4 fourth", src, out, stamps, nblk);
    CHECK(hipMemset(src, 0, h.size() * 2));
    run<1>("11 ds_read_b128 per 28 MFMAs, ZERO operands", src, out, stamps, nblk);
    return 0;
}
