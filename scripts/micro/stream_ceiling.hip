// stream_ceiling.hip -- what this chip streams, in the access patterns of this repository's kernels.
//
//   hipcc --offload-arch=gfx950 -O3 -o scripts/micro/_bin/stream_ceiling scripts/micro/stream_ceiling.hip
//   scripts/micro/_bin/stream_ceiling time    > profiles/r05_stream_ceiling.jsonl      (HIP-event timings, JSON lines)
//   rocprofv3 --pmc FETCH_SIZE -- scripts/micro/_bin/stream_ceiling calib             (one launch per kernel, known bytes)
//
// `time`: plain float4 copy / read / write next to the GEMM-epilogue kernels' pattern -- workgroups that own 128-row tiles of
// 1-KiB rows (a row = one wavefront access of 64 lanes x float4), R input tensors and W output tensors of n x 256 floats:
// forward epilogue 2 reads + 3 writes (X, A.X -> Zs, Zn, out), backward 4 reads + 2 writes -- as a tiled grid (2 260 workgroups
// for 289 k rows, as the kernels launch) and as a persistent grid (512 workgroups walking tiles), with default / non-temporal
// stores and loads.  If the plain copy reaches the guide's ~6.3 TB/s and the tiled pattern does not, the gap is the pattern.
// `calib`: kernels with KNOWN byte counts in the access shapes the PMC tables of profiles/ are read for (wide streaming,
// 1-KiB row gather, 400-B row gather at a 400-B pitch, dword gather, 256-B segment gather): FETCH_SIZE / WRITE_SIZE per kernel
// name against the printed bytes gives the correction factor per access shape (scripts/micro/calib_report.py).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <numeric>
#include <random>
#include <string>
#include <vector>

#define CK(x)                                                                          \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));        \
      exit(1);                                                                         \
    }                                                                                  \
  } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

template <int NT>
__device__ __forceinline__ f4 ld(const f4 *p) {
  if (NT) return __builtin_nontemporal_load(p);
  return *p;
}
template <int NT>
__device__ __forceinline__ void st(f4 *p, f4 v) {
  if (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

// ---- plain grid-stride kernels over n4 float4s
template <int NTL, int NTS>
__global__ void __launch_bounds__(256) copy_f4(const f4 *__restrict__ a, f4 *__restrict__ o, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) st<NTS>(o + i, ld<NTL>(a + i));
}
// 4 independent loads in flight per thread
template <int NTL, int NTS>
__global__ void __launch_bounds__(256) copy_f4_x4(const f4 *__restrict__ a, f4 *__restrict__ o, size_t n4) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    f4 v0 = ld<NTL>(a + i), v1 = ld<NTL>(a + i + stride), v2 = ld<NTL>(a + i + 2 * stride), v3 = ld<NTL>(a + i + 3 * stride);
    st<NTS>(o + i, v0); st<NTS>(o + i + stride, v1); st<NTS>(o + i + 2 * stride, v2); st<NTS>(o + i + 3 * stride, v3);
  }
  for (; i < n4; i += stride) st<NTS>(o + i, ld<NTL>(a + i));
}
__global__ void __launch_bounds__(256) read_f4(const f4 *__restrict__ a, float *__restrict__ o, size_t n4) {
  f4 acc = {0, 0, 0, 0};
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    f4 v0 = a[i], v1 = a[i + stride], v2 = a[i + 2 * stride], v3 = a[i + 3 * stride];
    acc += v0 + v1 + v2 + v3;
  }
  for (; i < n4; i += stride) acc += a[i];
  if (acc.x + acc.y + acc.z + acc.w == 1.2345f) o[0] = 1.f;
}
template <int NTS>
__global__ void __launch_bounds__(256) write_f4(f4 *__restrict__ o, size_t n4) {
  const f4 v = {1.f, 2.f, 3.f, 4.f};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) st<NTS>(o + i, v);
}

// ---- the GEMM-epilogue kernels' pattern: tiles of TR rows of 256 floats; wavefront w of the 4 takes rows w, w + 4, ...; per row
//      R loads (one per input tensor) and W stores (one per output tensor), two rows in flight.  PERSIST: grid-stride over tiles.
struct Ptrs {
  const f4 *in[4];
  f4 *out[3];
};
template <int R, int W, int NTL, int NTS, int TR, int PERSIST>
__global__ void __launch_bounds__(256, 2) tiles_rw(Ptrs p, uint32_t n) {
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  const uint32_t tiles = (n + TR - 1) / TR;
  for (uint32_t t = blockIdx.x; t < tiles; t += PERSIST ? gridDim.x : tiles) {
    const uint32_t r0 = t * TR, r1 = min(n, r0 + TR);
    for (uint32_t r = r0 + wv; r < r1; r += 8) {
      const uint32_t rb = r + 4;
      f4 a[R], b[R];
#pragma unroll
      for (int k = 0; k < R; ++k) a[k] = ld<NTL>(p.in[k] + (size_t)r * 64 + lane);
      if (rb < r1) {
#pragma unroll
        for (int k = 0; k < R; ++k) b[k] = ld<NTL>(p.in[k] + (size_t)rb * 64 + lane);
      }
      f4 s = a[0];
#pragma unroll
      for (int k = 1; k < R; ++k) s += a[k];
#pragma unroll
      for (int k = 0; k < W; ++k) st<NTS>(p.out[k] + (size_t)r * 64 + lane, s * (float)(k + 1));
      if (rb < r1) {
        f4 s2 = b[0];
#pragma unroll
        for (int k = 1; k < R; ++k) s2 += b[k];
#pragma unroll
        for (int k = 0; k < W; ++k) st<NTS>(p.out[k] + (size_t)rb * 64 + lane, s2 * (float)(k + 1));
      }
    }
  }
}

// ---- calibration kernels (known bytes)
// rows of ROWB bytes (multiple of 16) from a table of pitch PITCH bytes, picked by idx, written compact at pitch OPITCH; a row on
// LPR lanes
template <int ROWB, int PITCH, int OPITCH, int LPR>
__global__ void __launch_bounds__(256) calib_gather_rows(const char *__restrict__ table, const uint32_t *__restrict__ idx, char *__restrict__ out, uint32_t m) {
  const uint32_t sub = threadIdx.x % LPR;
  for (uint64_t r = ((uint64_t)blockIdx.x * 256 + threadIdx.x) / LPR; r < m; r += (uint64_t)gridDim.x * 256 / LPR) {
    const uint64_t src = (uint64_t)idx[r] * PITCH;
    for (uint32_t c = sub * 16; c < ROWB; c += LPR * 16)
      *reinterpret_cast<f4 *>(out + r * OPITCH + c) = *reinterpret_cast<const f4 *>(table + src + c);
  }
}
__global__ void __launch_bounds__(256) calib_gather_dword(const uint32_t *__restrict__ table, const uint32_t *__restrict__ idx, uint32_t *__restrict__ out, uint32_t m) {
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (uint64_t)gridDim.x * 256) out[i] = table[idx[i]];
}

static double time_ms(hipStream_t s, int iters, const std::function<void()> &launch) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) launch();
  CK(hipStreamSynchronize(s));
  CK(hipEventRecord(e0, s));
  for (int i = 0; i < iters; ++i) launch();
  CK(hipEventRecord(e1, s));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGetLastError());
  return ms / iters;
}

static void report(const char *name, double bytes, double ms, const char *note) {
  printf("{\"kernel\": \"%s\", \"bytes\": %.0f, \"us\": %.2f, \"GBps\": %.1f, \"frac_of_8TBps\": %.4f, \"note\": \"%s\"}\n", name, bytes, ms * 1e3,
         bytes / 1e9 / (ms / 1e3), bytes / 1e9 / (ms / 1e3) / 8000.0, note);
  fflush(stdout);
}

template <int R, int W, int NTL, int NTS, int TR, int PERSIST>
static void run_tiles(hipStream_t s, Ptrs p, uint32_t n, uint32_t grid_persist, const char *name) {
  const uint32_t tiles = (n + TR - 1) / TR;
  const uint32_t grid = PERSIST ? std::min(grid_persist, tiles) : tiles;
  double ms = time_ms(s, 20, [&] { hipLaunchKernelGGL((tiles_rw<R, W, NTL, NTS, TR, PERSIST>), dim3(grid), dim3(256), 0, s, p, n); });
  char note[160];
  snprintf(note, sizeof note, "%d reads + %d writes of n x 1 KiB rows, n = %u, %d-row tiles, %s grid of %u, loads %s, stores %s", R, W, n, TR,
           PERSIST ? "persistent" : "tiled", grid, NTL ? "nt" : "default", NTS ? "nt" : "default");
  report(name, (double)(R + W) * n * 1024.0, ms, note);
}

int main(int argc, char **argv) {
  const std::string mode = argc > 1 ? argv[1] : "time";
  hipStream_t s;
  CK(hipStreamCreate(&s));
  if (mode == "time") {
    const uint32_t n = 289308;                          // rows of the benchmark batch (289 k x 256 floats = 296 MB per tensor)
    const size_t n4 = (size_t)n * 64;
    f4 *buf[7];
    for (auto &b : buf) { CK(hipMalloc(&b, n4 * 16)); CK(hipMemsetAsync(b, 0, n4 * 16, s)); }
    float *small; CK(hipMalloc(&small, 256));
    // plain streaming, two sizes (296 MB: one tensor of the step, above the 256 MB Infinity Cache together with its copy; 1.18 GB)
    for (int big = 0; big < 2; ++big) {
      f4 *src = buf[0], *dst = buf[1];
      size_t m4 = n4;
      f4 *bs = nullptr, *bd = nullptr;
      if (big) { m4 = n4 * 4; CK(hipMalloc(&bs, m4 * 16)); CK(hipMalloc(&bd, m4 * 16)); CK(hipMemsetAsync(bs, 0, m4 * 16, s)); src = bs; dst = bd; }
      const char *sz = big ? "1.18 GB per tensor" : "296 MB per tensor";
      for (uint32_t grid : {512u, 2048u, 8192u, 65536u}) {
        char nm[64], note[128];
        snprintf(nm, sizeof nm, "copy_f4_g%u%s", grid, big ? "_big" : "");
        snprintf(note, sizeof note, "float4 grid-stride copy, %s, grid %u x 256", sz, grid);
        report(nm, 2.0 * m4 * 16, time_ms(s, 20, [&] { hipLaunchKernelGGL((copy_f4<0, 0>), dim3(grid), dim3(256), 0, s, src, dst, m4); }), note);
      }
      {
        char nm[64], note[128];
        snprintf(nm, sizeof nm, "copy_f4_x4_g2048%s", big ? "_big" : "");
        snprintf(note, sizeof note, "float4 copy, 4 loads in flight per thread, %s", sz);
        report(nm, 2.0 * m4 * 16, time_ms(s, 20, [&] { hipLaunchKernelGGL((copy_f4_x4<0, 0>), dim3(2048), dim3(256), 0, s, src, dst, m4); }), note);
        snprintf(nm, sizeof nm, "copy_f4_x4_nt_g2048%s", big ? "_big" : "");
        snprintf(note, sizeof note, "... with non-temporal loads and stores, %s", sz);
        report(nm, 2.0 * m4 * 16, time_ms(s, 20, [&] { hipLaunchKernelGGL((copy_f4_x4<1, 1>), dim3(2048), dim3(256), 0, s, src, dst, m4); }), note);
        snprintf(nm, sizeof nm, "copy_f4_x4_nts_g2048%s", big ? "_big" : "");
        snprintf(note, sizeof note, "... with non-temporal stores only, %s", sz);
        report(nm, 2.0 * m4 * 16, time_ms(s, 20, [&] { hipLaunchKernelGGL((copy_f4_x4<0, 1>), dim3(2048), dim3(256), 0, s, src, dst, m4); }), note);
        snprintf(nm, sizeof nm, "read_f4_g2048%s", big ? "_big" : "");
        snprintf(note, sizeof note, "read-only float4 stream, %s", sz);
        report(nm, 1.0 * m4 * 16, time_ms(s, 20, [&] { hipLaunchKernelGGL(read_f4, dim3(2048), dim3(256), 0, s, src, small, m4); }), note);
        snprintf(nm, sizeof nm, "write_f4_g2048%s", big ? "_big" : "");
        snprintf(note, sizeof note, "write-only float4 stream, %s", sz);
        report(nm, 1.0 * m4 * 16, time_ms(s, 20, [&] { hipLaunchKernelGGL((write_f4<0>), dim3(2048), dim3(256), 0, s, dst, m4); }), note);
        snprintf(nm, sizeof nm, "write_f4_nt_g2048%s", big ? "_big" : "");
        snprintf(note, sizeof note, "write-only float4 stream, non-temporal stores, %s", sz);
        report(nm, 1.0 * m4 * 16, time_ms(s, 20, [&] { hipLaunchKernelGGL((write_f4<1>), dim3(2048), dim3(256), 0, s, dst, m4); }), note);
        hipMemcpyAsync(dst, src, m4 * 16, hipMemcpyDeviceToDevice, s);
        report(big ? "hipMemcpyDtoD_big" : "hipMemcpyDtoD", 2.0 * m4 * 16, time_ms(s, 20, [&] { hipMemcpyAsync(dst, src, m4 * 16, hipMemcpyDeviceToDevice, s); }), sz);
      }
      if (big) { CK(hipFree(bs)); CK(hipFree(bd)); }
    }
    Ptrs p;
    for (int k = 0; k < 4; ++k) p.in[k] = buf[k];
    for (int k = 0; k < 3; ++k) p.out[k] = buf[4 + k];
    // forward GEMM-epilogue pattern: 2 reads + 3 writes
    run_tiles<2, 3, 0, 0, 128, 0>(s, p, n, 512, "fwd_2r3w_tiled128");
    run_tiles<2, 3, 0, 1, 128, 0>(s, p, n, 512, "fwd_2r3w_tiled128_nts");
    run_tiles<2, 3, 1, 1, 128, 0>(s, p, n, 512, "fwd_2r3w_tiled128_ntls");
    run_tiles<2, 3, 0, 0, 128, 1>(s, p, n, 512, "fwd_2r3w_persist128");
    run_tiles<2, 3, 0, 0, 32, 1>(s, p, n, 512, "fwd_2r3w_persist32");
    run_tiles<2, 3, 0, 1, 32, 1>(s, p, n, 512, "fwd_2r3w_persist32_nts");
    run_tiles<2, 3, 0, 0, 32, 1>(s, p, n, 1024, "fwd_2r3w_persist32_g1024");
    run_tiles<2, 3, 0, 0, 32, 1>(s, p, n, 2048, "fwd_2r3w_persist32_g2048");
    run_tiles<2, 3, 0, 0, 8, 1>(s, p, n, 2048, "fwd_2r3w_persist8_g2048");
    // the tensors the forward MUST move (X, A.X in; out out)
    run_tiles<2, 1, 0, 0, 128, 0>(s, p, n, 512, "fwd_2r1w_tiled128");
    // backward GEMM-epilogue pattern: 4 reads + 2 writes
    run_tiles<4, 2, 0, 0, 128, 0>(s, p, n, 512, "bwd_4r2w_tiled128");
    run_tiles<4, 2, 0, 1, 128, 0>(s, p, n, 512, "bwd_4r2w_tiled128_nts");
    run_tiles<4, 2, 0, 0, 32, 1>(s, p, n, 1024, "bwd_4r2w_persist32_g1024");
    // SpMM-like and weight-gradient-like mixes
    run_tiles<1, 1, 0, 0, 128, 0>(s, p, n, 512, "rw_1r1w_tiled128");
    run_tiles<3, 1, 0, 0, 128, 0>(s, p, n, 512, "rw_3r1w_tiled128");
    run_tiles<3, 1, 0, 0, 32, 1>(s, p, n, 2048, "rw_3r1w_persist32_g2048");
    return 0;
  }
  if (mode == "calib") {
    // known-byte kernels, ONE launch each (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE per kernel name)
    const size_t GiB = 1ull << 30;
    char *table, *out;
    CK(hipMalloc(&table, 2 * GiB + 4096)); CK(hipMalloc(&out, GiB + 4096));
    CK(hipMemset(table, 1, 2 * GiB)); CK(hipMemset(out, 0, GiB));
    std::mt19937_64 rng(5);
    auto upload_perm_prefix = [&](uint32_t universe, uint32_t m) {
      std::vector<uint32_t> v(universe);
      std::iota(v.begin(), v.end(), 0u);
      for (uint32_t i = 0; i < m; ++i) std::swap(v[i], v[i + rng() % (universe - i)]);
      uint32_t *d;
      CK(hipMalloc(&d, (size_t)m * 4));
      CK(hipMemcpy(d, v.data(), (size_t)m * 4, hipMemcpyHostToDevice));
      return std::make_pair(d, std::vector<uint32_t>(v.begin(), v.begin() + m));
    };
    auto lines = [](const std::vector<uint32_t> &idx, uint64_t pitch, uint64_t rowb, uint64_t gran) {
      double t = 0;
      for (uint32_t i : idx) t += (double)(((uint64_t)i * pitch + rowb - 1) / gran - ((uint64_t)i * pitch) / gran + 1);
      return t * gran;
    };
    CK(hipDeviceSynchronize());
    // 1. wide streaming: copy / read / write of 1 GiB
    hipLaunchKernelGGL((copy_f4_x4<0, 0>), dim3(2048), dim3(256), 0, s, (const f4 *)table, (f4 *)out, GiB / 16);
    printf("{\"calib\": \"copy_f4_x4\", \"read_bytes\": %.0f, \"write_bytes\": %.0f}\n", (double)GiB, (double)GiB);
    hipLaunchKernelGGL(read_f4, dim3(2048), dim3(256), 0, s, (const f4 *)table, (float *)out, GiB / 16);
    printf("{\"calib\": \"read_f4\", \"read_bytes\": %.0f, \"write_bytes\": 0}\n", (double)GiB);
    hipLaunchKernelGGL((write_f4<0>), dim3(2048), dim3(256), 0, s, (f4 *)out, GiB / 16);
    printf("{\"calib\": \"write_f4\", \"read_bytes\": 0, \"write_bytes\": %.0f}\n", (double)GiB);
    CK(hipStreamSynchronize(s));
    // 2. 1-KiB rows gathered from 2 Mi rows (2 GiB), 256 Ki distinct rows -> compact
    {
      const uint32_t m = 262144;
      auto [d, h] = upload_perm_prefix(2097152u, m);
      hipLaunchKernelGGL((calib_gather_rows<1024, 1024, 1024, 64>), dim3(4096), dim3(256), 0, s, table, d, out, m);
      CK(hipStreamSynchronize(s));
      printf("{\"calib\": \"calib_gather_rows<1024, 1024, 1024, 64>\", \"shape\": \"1-KiB rows, line aligned\", \"read_bytes\": %.0f, \"index_bytes\": %.0f, \"write_bytes\": %.0f}\n",
             (double)m * 1024, (double)m * 4, (double)m * 1024);
      CK(hipFree(d));
    }
    // 3. 400-B rows at a 400-B pitch (the feature table of gather_F100), 512 Ki distinct rows -> 512-B pitch (400 B written per row)
    {
      const uint32_t m = 524288, universe = (uint32_t)((2 * GiB) / 400);
      auto [d, h] = upload_perm_prefix(universe, m);
      hipLaunchKernelGGL((calib_gather_rows<400, 400, 512, 32>), dim3(4096), dim3(256), 0, s, table, d, out, m);
      CK(hipStreamSynchronize(s));
      printf("{\"calib\": \"calib_gather_rows<400, 400, 512, 32>\", \"shape\": \"400-B rows at a 400-B pitch\", \"read_bytes\": %.0f, \"read_bytes_64B_lines\": %.0f, "
             "\"read_bytes_128B_lines\": %.0f, \"index_bytes\": %.0f, \"write_bytes\": %.0f, \"write_bytes_64B_lines\": %.0f, \"write_bytes_128B_lines\": %.0f}\n",
             (double)m * 400, lines(h, 400, 400, 64), lines(h, 400, 400, 128), (double)m * 4, (double)m * 400, (double)m * 448, (double)m * 512);
      CK(hipFree(d));
    }
    // 4. dwords gathered from 512 Mi dwords (2 GiB), 16 Mi distinct -> compact
    {
      const uint32_t m = 16777216;
      auto [d, h] = upload_perm_prefix(536870912u, m);
      hipLaunchKernelGGL(calib_gather_dword, dim3(8192), dim3(256), 0, s, (const uint32_t *)table, d, (uint32_t *)out, m);
      CK(hipStreamSynchronize(s));
      printf("{\"calib\": \"calib_gather_dword\", \"shape\": \"4-B reads at random positions\", \"read_bytes\": %.0f, \"read_bytes_64B_lines\": %.0f, \"read_bytes_128B_lines\": %.0f, "
             "\"index_bytes\": %.0f, \"write_bytes\": %.0f}\n",
             (double)m * 4, (double)m * 64, (double)m * 128, (double)m * 4, (double)m * 4);
      CK(hipFree(d));
    }
    // 5. 256-B segments at 16-B aligned random positions (a CSR row of ~64 ids as the sampler's scan reads it), 2 Mi segments
    {
      const uint32_t m = 2097152, universe = (uint32_t)((2 * GiB) / 272);
      auto [d, h] = upload_perm_prefix(universe, m);
      hipLaunchKernelGGL((calib_gather_rows<256, 272, 256, 16>), dim3(4096), dim3(256), 0, s, table, d, out, m);
      CK(hipStreamSynchronize(s));
      printf("{\"calib\": \"calib_gather_rows<256, 272, 256, 16>\", \"shape\": \"256-B segments, 16-B aligned (CSR rows)\", \"read_bytes\": %.0f, \"read_bytes_64B_lines\": %.0f, "
             "\"read_bytes_128B_lines\": %.0f, \"index_bytes\": %.0f, \"write_bytes\": %.0f}\n",
             (double)m * 256, lines(h, 272, 256, 64), lines(h, 272, 256, 128), (double)m * 4, (double)m * 256);
      CK(hipFree(d));
    }
    CK(hipDeviceSynchronize());
    return 0;
  }
  fprintf(stderr, "usage: stream_ceiling time|calib\n");
  return 2;
}
