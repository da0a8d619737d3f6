// coresidency.hip — which resource keeps a kernel that is dispatched WHILE a resident grid runs off the chip?  (diagnostic, r04)
//
// Observation (tools/midsweep_probe.py): a tiny kernel launched 2 ms into the residual sweep finishes within 0.05-0.1 ms; ONE
// workgroup of k_dlt4 (4 waves, 72 registers, 78 KB of LDS) launched the same way only runs when the sweep ends — also when
// the sweep leaves a whole workgroup slot per compute unit free.  This microbenchmark separates the candidates: a HOG
// kernel shaped like the sweep (256 threads, ~88 registers, 1.5 KB of LDS, `per_cu` workgroups per CU, spinning for a few
// milliseconds) and, launched into it from another stream, a PROBE of one workgroup per CU with a chosen LDS size and
// register count that does a trivial amount of work.  Reported: the probe's completion time after its launch.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(88)))
k_hog(double* out, long long ticks)
{
    __shared__ double s[193];                      // 1.5 KB like the sweep
    double acc[40];
    for (int i = 0; i < 40; ++i) acc[i] = threadIdx.x * 1e-3 + i;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) {
#pragma unroll
        for (int i = 0; i < 40; ++i) acc[i] = acc[i] * 1.0000001 + 1e-9;
    }
    double a = 0;
    for (int i = 0; i < 40; ++i) a += acc[i];
    if (threadIdx.x < 193) s[threadIdx.x] = a;
    __syncthreads();
    if (a == 12345.678) out[blockIdx.x] = s[(threadIdx.x + 1) % 193];
}

template <int NV>
__global__ void __launch_bounds__(256)
k_probe(double* out, int lds_doubles)
{
    extern __shared__ double dyn[];
    double r[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) r[i] = threadIdx.x + i;
    for (int i = threadIdx.x; i < lds_doubles; i += 256) dyn[i] = i;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < NV; ++i) r[i] = r[i] * 1.0000001 + (lds_doubles > 0 ? dyn[(threadIdx.x + i) % lds_doubles] : 0.0);
    double a = 0;
#pragma unroll
    for (int i = 0; i < NV; ++i) a += r[i];
    if (a == 12345.678) out[blockIdx.x] = a;
}

int main(int argc, char** argv)
{
    int cus = 256;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    cus = prop.multiProcessorCount;
    double* out;
    CK(hipMalloc(&out, 1 << 20));
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    const long long hog_ticks = 500000;            // 5 ms at 100 MHz
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)k_hog, 256, 0));
    printf("%d CUs; the hog's occupancy by the query: %d workgroups per CU\n", cus, occ);
    CK(hipFuncSetAttribute((const void*)k_probe<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)k_probe<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)k_probe<56>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int lds_kb[] = { 0, 8, 16, 32, 48, 60, 64, 72, 78, 96, 128 };
    for (int per_cu : { 5, 4, 3 }) {
        for (int nv : { 8, 32, 56 }) {
            for (int kb : lds_kb) {
                double best = 1e9, ref = 1e9;
                for (int rep = 0; rep < 3; ++rep) {
                    for (int with_hog = 1; with_hog >= 0; --with_hog) {
                        CK(hipDeviceSynchronize());
                        if (with_hog) {
                            hipLaunchKernelGGL(k_hog, dim3(per_cu * cus), dim3(256), 0, a, out, hog_ticks);
                            std::this_thread::sleep_for(std::chrono::milliseconds(1));
                        }
                        const auto t0 = std::chrono::steady_clock::now();
                        const size_t lds = (size_t)kb * 1024;
                        if (nv == 8) hipLaunchKernelGGL(k_probe<8>, dim3(cus), dim3(256), lds, b, out, kb * 128);
                        else if (nv == 32) hipLaunchKernelGGL(k_probe<32>, dim3(cus), dim3(256), lds, b, out, kb * 128);
                        else hipLaunchKernelGGL(k_probe<56>, dim3(cus), dim3(256), lds, b, out, kb * 128);
                        CK(hipGetLastError());
                        CK(hipStreamSynchronize(b));
                        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                        if (with_hog) best = ms < best ? ms : best; else ref = ms < ref ? ms : ref;
                    }
                }
                printf("hog %d per CU | probe %3d doubles in registers per lane, %3d KB LDS: done %.3f ms after its launch beside the hog, %.3f ms alone%s\n",
                       per_cu, nv, kb, best, ref, best > 1.0 ? "   <-- waited for the hog" : "");
                fflush(stdout);
            }
        }
    }
    return 0;
}
