// How many workgroups of (threads, LDS bytes) run concurrently on MI355X?  Each block spins ~20 us and
// records (start, end, XCC id, CU id); concurrency = sum(block time) / kernel time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <set>
__global__ void spin(long long ticks, unsigned long long* out) {
    extern __shared__ char lds[];
    long long t0 = wall_clock64();
    if (threadIdx.x == 0) lds[0] = 1;
    __syncthreads();
    if (ticks < 0) ticks = 1000 + (long long)((blockIdx.x * 2654435761u) >> 20) % 3000;  // 10-40 us, per block
    while (wall_clock64() - t0 < ticks) { __builtin_amdgcn_s_sleep(8); }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[blockIdx.x * 4 + 0] = t0;
        out[blockIdx.x * 4 + 1] = wall_clock64();
        out[blockIdx.x * 4 + 2] = hw;
        out[blockIdx.x * 4 + 3] = xcc;
    }
}
int main(int argc, char** argv) {
    int blocks = argc > 1 ? atoi(argv[1]) : 1024, threads = argc > 2 ? atoi(argv[2]) : 1024, ldskb = argc > 3 ? atoi(argv[3]) : 152;
    unsigned long long* d;
    hipMalloc(&d, blocks * 32);
    hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, ldskb * 1024);
    for (int it = 0; it < 2; ++it) spin<<<blocks, threads, ldskb * 1024>>>(argc > 4 ? -1 : 2000, d);  // 20 us, or 10-40 us with a 4th argument
    hipError_t e = hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), d, blocks * 32, hipMemcpyDeviceToHost);
    unsigned long long mn = ~0ull, mx = 0, sum = 0;
    std::set<unsigned long long> cus;
    for (int i = 0; i < blocks; ++i) {
        mn = std::min(mn, h[i * 4]); mx = std::max(mx, h[i * 4 + 1]); sum += h[i * 4 + 1] - h[i * 4];
        unsigned hw = (unsigned)h[i * 4 + 2];
        cus.insert(((h[i * 4 + 3] & 0xF) << 16) | (hw & 0xFFF00));  // xcc | se/sh/cu bits (approx)
    }
    printf("%s blocks=%d threads=%d lds=%dKB kernel=%.1f us concurrency=%.1f distinct_cu_keys=%zu\n", hipGetErrorString(e), blocks, threads,
           ldskb, (mx - mn) / 100.0, (double)sum / (mx - mn), cus.size());
    return 0;
}
