// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on this GPU for the access widths the kernels here use:
// streams a buffer larger than the 256 MiB Infinity Cache with 4-byte and 16-byte loads / stores per lane.
// Build: hipcc -O3 --offload-arch=gfx950 -o fetch_calib tools/fetch_calib.hip
// Run:   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- ./fetch_calib     (and again with WRITE_SIZE)
// Every kernel moves exactly BYTES bytes; the counters (KiB) divided by BYTES / 1024 give the factor.
#include <hip/hip_runtime.h>
#include <cstdio>
static const size_t BYTES = 1ull << 30;
__global__ void k_read4(const int* __restrict__ a, size_t n, int* out) {
    int s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += a[i];
    if (s == 0x12345678) *out = s;
}
__global__ void k_read16(const int4* __restrict__ a, size_t n, int* out) {
    int s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { int4 v = a[i]; s += v.x ^ v.y ^ v.z ^ v.w; }
    if (s == 0x12345678) *out = s;
}
// 4-byte loads of 256-byte row pieces, one piece per wave and step, pieces a "plane" apart (the access of the EDT passes)
__global__ void k_read4_rows(const int* __restrict__ a, size_t row_ints, size_t rows, int* out) {
    int s = 0;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63, waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    const size_t pieces = row_ints / 64;
    for (size_t p = wave; p < pieces; p += waves)
        for (size_t r = 0; r < rows; r++) s += a[r * row_ints + p * 64 + lane];
    if (s == 0x12345678) *out = s;
}
__global__ void k_write4(int* __restrict__ a, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = (int)i;
}
__global__ void k_write16(int4* __restrict__ a, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = make_int4((int)i, 1, 2, 3);
}
int main() {
    int *a, *out;
    if (hipMalloc(&a, BYTES) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(a, 0, BYTES);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; rep++) {
        k_read4<<<4096, 256>>>(a, BYTES / 4, out);
        k_read16<<<4096, 256>>>((const int4*)a, BYTES / 16, out);
        k_read4_rows<<<4096, 256>>>(a, BYTES / 4 / 128, 128, out);
        k_write4<<<4096, 256>>>(a, BYTES / 4);
        k_write16<<<4096, 256>>>((int4*)a, BYTES / 16);
    }
    hipError_t e = hipDeviceSynchronize();
    printf("%s\n", hipGetErrorString(e));
    return e != hipSuccess;
}
