// What a cross-wavefront hand-off inside one workgroup costs on gfx950 -- the price of splitting ONE sequence's solve over several
// wavefronts.  A workgroup of W wavefronts (one per SIMD for W <= 4) runs `rounds` rounds; in each round
// wavefront (round % W) does `work` dependent float64 FMAs on a value it reads from LDS, writes the result back, and every wavefront
// meets at s_barrier.  With work = 0 a round is the bare hand-off: ds_write -> s_barrier -> ds_read.  The same chain run by ONE
// wavefront without barriers is the reference.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/handoff tools/handoff_probe.hip && /tmp/handoff
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double *o, int rounds, int work, int split) {
  __shared__ double box[64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, W = blockDim.x >> 6;
  if (wave == 0) box[lane] = 1.0 + lane * 1e-3;
  __syncthreads();
  double v = 0.0;
  for (int r = 0; r < rounds; ++r) {
    if (!split || wave == r % W) {
      v = box[lane];
      for (int i = 0; i < work; ++i) v = fma(v, 1.0000001, 1e-9);
      box[lane] = v;
    }
    if (split) __syncthreads();
  }
  if (wave == 0) o[blockIdx.x * 64 + lane] = v + box[lane];
}
static float run(int waves, int rounds, int work, int split, double *d) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  k<<<1, 64 * waves>>>(d, 8, work, split);
  hipEventRecord(a);
  k<<<1, 64 * waves>>>(d, rounds, work, split);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms;
}
int main() {
  double *d;
  hipMalloc(&d, 64 * 8);
  const int rounds = 20000;
  for (int work : {0, 32, 128}) {
    const float one = run(1, rounds, work, 0, d);
    printf("work %3d FMAs/round: one wavefront, no barrier %.1f ns/round", work, one * 1e6 / rounds);
    for (int w : {2, 4}) printf(" | %d wavefronts taking turns %.1f ns/round", w, run(w, rounds, work, 1, d) * 1e6 / rounds);
    printf("\n");
  }
  return 0;
}
