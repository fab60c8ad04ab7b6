// How fast can the stem kernels read their image tiles?  A patch of a 4480 x 4480 fp32 image is read as row segments of
// (2 * TX + 4) floats out of 17.9 KB-stride rows (forward stem: 36 rows x 68 floats per tile; stem weight gradient: 20 rows
// x 68 floats).  This benchmark streams the 448 x 448 patches of 64 images with the same tile decomposition and segment
// lengths 68 / 132 / 260 / 452 floats, one workgroup per tile, and reports GB/s of the bytes requested.
//   build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/imgreadbench.hip -o tools/imgreadbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void read_tiles(const float* __restrict__ img, const int* __restrict__ pos, int W, int P, int rows, int seg, int tiles_x,
                           int tiles_y, float* __restrict__ sink) {
  const int n = blockIdx.y, t = blockIdx.x, ty = t / tiles_x, tx = t % tiles_x;
  const long long plane = (long long)W * W;
  const float* base = img + (long long)n * 3 * plane + (long long)pos[2 * n] * P * W + (long long)pos[2 * n + 1] * P;
  const int y0 = ty * (rows - 4), x0 = tx * (seg - 4);
  float acc = 0.0f;
  const int total = 3 * rows * seg;
  for (int i = threadIdx.x; i < total; i += 256) {
    const int c = i / (rows * seg), r = (i / seg) % rows, q = i % seg;
    const int iy = y0 + r, ix = x0 + q;
    if (iy < P && ix < P) acc += base[c * plane + (long long)iy * W + ix];
  }
  if (acc == 12345.678f) sink[0] = acc;
}

// the same tiles with 16-byte loads (segment start and length multiples of 4 floats) and all loads of a thread issued
// before the first use
template <int NL>
__global__ void read_tiles4(const float* __restrict__ img, const int* __restrict__ pos, int W, int P, int rows, int seg, int tiles_x,
                            int tiles_y, float* __restrict__ sink) {
  const int n = blockIdx.y, t = blockIdx.x, ty = t / tiles_x, tx = t % tiles_x;
  const long long plane = (long long)W * W;
  const float* base = img + (long long)n * 3 * plane + (long long)pos[2 * n] * P * W + (long long)pos[2 * n + 1] * P;
  const int y0 = ty * (rows - 4), x0 = tx * (seg - 4), sq = seg / 4;
  const int total = 3 * rows * sq;
  float4 v[NL];
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    int i = threadIdx.x + 256 * j;
    i = i < total ? i : total - 1;
    const int c = i / (rows * sq), r = (i / sq) % rows, q = i % sq;
    int iy = y0 + r, ix = x0 + 4 * q;
    iy = iy < P ? iy : P - 1; ix = ix < P - 3 ? ix : P - 4;
    v[j] = *reinterpret_cast<const float4*>(base + c * plane + (long long)iy * W + ix);
  }
  float acc = 0.0f;
#pragma unroll
  for (int j = 0; j < NL; ++j) acc += v[j].x + v[j].y + v[j].z + v[j].w;
  if (acc == 12345.678f) sink[0] = acc;
}

int main() {
  const int N = 64, W = 4480, P = 448;
  float* img; int* pos; float* sink;
  CK(hipMalloc(&img, (size_t)N * 3 * W * W * 4)); CK(hipMalloc(&pos, N * 2 * 4)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(img, 0, (size_t)N * 3 * W * W * 4));
  int hp[2 * N];
  for (int i = 0; i < N; ++i) { hp[2 * i] = (i * 7) % 10; hp[2 * i + 1] = (i * 3) % 10; }
  CK(hipMemcpy(pos, hp, sizeof(hp), hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  struct Cfg { int rows, seg; };
  for (Cfg c : {Cfg{36, 68}, Cfg{20, 68}, Cfg{20, 132}, Cfg{12, 132}, Cfg{12, 260}, Cfg{36, 132}, Cfg{20, 452}, Cfg{68, 452}}) {
    const int tiles_x = (P + c.seg - 5) / (c.seg - 4), tiles_y = (P + c.rows - 5) / (c.rows - 4);
    dim3 grid(tiles_x * tiles_y, N);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(read_tiles, grid, dim3(256), 0, nullptr, img, pos, W, P, c.rows, c.seg, tiles_x, tiles_y, sink);
    CK(hipEventRecord(e0));
    const int it = 10;
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL(read_tiles, grid, dim3(256), 0, nullptr, img, pos, W, P, c.rows, c.seg, tiles_x, tiles_y, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)N * tiles_x * tiles_y * 3.0 * c.rows * c.seg * 4.0;
    printf("tile %2d rows x %3d floats (%4d B segments), %4d tiles/patch: %7.1f us, %6.0f GB/s requested, %6.0f GB/s of the patch bytes", c.rows, c.seg,
           c.seg * 4, tiles_x * tiles_y, ms * 1e3 / it, bytes / (ms * 1e-3 / it) / 1e9, (double)N * 3 * P * P * 4 / (ms * 1e-3 / it) / 1e9);
    const int nl = (3 * c.rows * (c.seg / 4) + 255) / 256;
    if (nl <= 8) {
      auto k4 = read_tiles4<8>;
      for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k4, grid, dim3(256), 0, nullptr, img, pos, W, P, c.rows, c.seg, tiles_x, tiles_y, sink);
      CK(hipEventRecord(e0));
      for (int i = 0; i < it; ++i) hipLaunchKernelGGL(k4, grid, dim3(256), 0, nullptr, img, pos, W, P, c.rows, c.seg, tiles_x, tiles_y, sink);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf(" | 16-byte loads, all in flight: %7.1f us, %6.0f GB/s requested", ms * 1e3 / it, bytes / (ms * 1e-3 / it) / 1e9);
    }
    printf("\n");
  }
  return 0;
}
