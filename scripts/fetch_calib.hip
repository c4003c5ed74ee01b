// fetch_calib.hip -- what does rocprofv3's FETCH_SIZE report for the access shapes of the HBM megakernel?
//
// MI355X_MICROARCH.md: "On gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read (16 B/lane)
// ... other access widths are uncalibrated: calibrate on a known byte count in your own access pattern".  The C4 kernel's
// loads are NOT coalesced streams: every lane reads one 64-byte quantised node (four global_load_dwordx4 off one per-lane
// pointer) or one 48-byte triangle record at an effectively random address.  This program issues a known number of bytes
// in that shape (and in the two streaming shapes for reference) over a 2 GiB buffer -- far beyond the 256 MiB Infinity
// Cache, every 128-byte line touched at most once -- so that bytes requested == bytes that must come from HBM.
//
//   hipcc --offload-arch=gfx950 -O3 scripts/fetch_calib.hip -o /tmp/fetch_calib
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- /tmp/fetch_calib
//
// Kernels (each launched once after a warm-up of the same kind on a different region):
//   stream16        : 16 B per lane, consecutive lanes consecutive addresses (the guide's calibrated case)
//   stream4         : 4 B per lane coalesced (the megakernel's per-pixel offset read)
//   node64_random   : per lane a random 64-byte-aligned node, 4 x 16-byte loads (the 4-wide BVH node fetch)
//   node64_pairs    : the same, but lanes 2k and 2k+1 read the two nodes of one 128-byte line (siblings are adjacent)
//   tri48_random    : per lane a random 48-byte record, 3 x 16-byte loads (triangle records)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

__device__ inline uint32_t pcg(uint32_t v) {
    uint32_t s = v * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    return (w >> 22u) ^ w;
}

__global__ void stream16(const v4f *p, size_t n16, float *sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    v4f acc = {0, 0, 0, 0};
    for (; i < n16; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc.x == 12345.0f) *sink = acc.y + acc.z + acc.w;
}
__global__ void stream4(const float *p, size_t n4, float *sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0;
    for (; i < n4; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 12345.0f) *sink = acc;
}
// a bijection of [0, n) for n a power of two: odd multiplier + xor-shift keeps every line touched exactly once
__device__ inline uint32_t scatter(uint32_t i, uint32_t mask) {
    uint32_t x = (i * 2654435761u) & mask;
    x ^= (x >> 7);
    x = (x * 40503u + 12345u) & mask;  // odd multiplier mod 2^k is a bijection; xor-shift right is too
    return x & mask;
}
__global__ void node64_random(const v4f *p, uint32_t nodes_mask, uint32_t per_thread, float *sink) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    v4f acc = {0, 0, 0, 0};
    for (uint32_t k = 0; k < per_thread; ++k) {
        const uint32_t node = scatter(t + k * nt, nodes_mask);
        const v4f *q = p + (size_t)node * 4;
        acc += q[0] + q[1] + q[2] + q[3];
    }
    if (acc.x == 12345.0f) *sink = acc.y;
}
__global__ void node64_pairs(const v4f *p, uint32_t nodes_mask, uint32_t per_thread, float *sink) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    v4f acc = {0, 0, 0, 0};
    for (uint32_t k = 0; k < per_thread; ++k) {
        const uint32_t pair = scatter((t >> 1) + k * (nt >> 1), nodes_mask >> 1);
        const v4f *q = p + ((size_t)pair * 2 + (t & 1u)) * 4;
        acc += q[0] + q[1] + q[2] + q[3];
    }
    if (acc.x == 12345.0f) *sink = acc.y;
}
__global__ void tri48_random(const v4f *p, uint32_t recs_mask, uint32_t per_thread, float *sink) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    v4f acc = {0, 0, 0, 0};
    for (uint32_t k = 0; k < per_thread; ++k) {
        const uint32_t rec = scatter(t + k * nt, recs_mask);
        const v4f *q = p + (size_t)rec * 3;
        acc += q[0] + q[1] + q[2];
    }
    if (acc.x == 12345.0f) *sink = acc.y;
}

int main() {
    const size_t bytes = 2ull << 30;  // 2 GiB
    void *buf; float *sink;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMalloc((void **)&sink, 4));
    CHECK(hipMemset(buf, 0, bytes));
    CHECK(hipDeviceSynchronize());
    const dim3 grid(256 * 8), block(256);
    const uint32_t nt = grid.x * block.x;                 // 524,288 threads
    // every kernel reads each of its bytes once; between kernels the 2 GiB memset evicts the caches
    auto flush = [&]() { CHECK(hipMemset(buf, 0, bytes)); CHECK(hipDeviceSynchronize()); };

    flush();
    stream16<<<grid, block>>>((const v4f *)buf, bytes / 16, sink);
    CHECK(hipDeviceSynchronize());
    printf("stream16       requested_bytes %zu\n", bytes);

    flush();
    stream4<<<grid, block>>>((const float *)buf, bytes / 8 / 4, sink);  // 256 MiB worth of dwords
    CHECK(hipDeviceSynchronize());
    printf("stream4        requested_bytes %zu\n", bytes / 8);

    flush();
    const uint32_t nodes = (uint32_t)(bytes / 64);        // 2^25 nodes
    const uint32_t per_thread_n = 16;                     // 8.4 M nodes = 512 MiB, each node once
    node64_random<<<grid, block>>>((const v4f *)buf, nodes - 1, per_thread_n, sink);
    CHECK(hipDeviceSynchronize());
    printf("node64_random  requested_bytes %zu  (lines touched: %zu x 128 B = %zu if a 64-B node costs its whole line)\n",
           (size_t)nt * per_thread_n * 64, (size_t)nt * per_thread_n, (size_t)nt * per_thread_n * 128);

    flush();
    node64_pairs<<<grid, block>>>((const v4f *)buf, nodes - 1, per_thread_n, sink);
    CHECK(hipDeviceSynchronize());
    printf("node64_pairs   requested_bytes %zu\n", (size_t)nt * per_thread_n * 64);

    flush();
    const uint32_t recs_pow2 = 1u << 25;                  // 2^25 records x 48 B = 1.5 GiB
    tri48_random<<<grid, block>>>((const v4f *)buf, recs_pow2 - 1, per_thread_n, sink);
    CHECK(hipDeviceSynchronize());
    printf("tri48_random   requested_bytes %zu\n", (size_t)nt * per_thread_n * 48);
    CHECK(hipFree(buf)); CHECK(hipFree(sink));
    return 0;
}
