// micro-benchmark: cost of a barrier among W workgroups implemented with a global atomic counter + __threadfence()
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ void bar_kernel(int* ctr, float* data, int W, int iters, int stride, int payload) {
  // participating workgroups: blockIdx.x % stride == 0 (stride 8 pins them to one XCD), first W of them
  const int id = blockIdx.x / stride;
  if (blockIdx.x % stride != 0 || id >= W) return;
  float acc = 0.0f;
  for (int it = 0; it < iters; ++it) {
    // payload: each workgroup writes `payload` floats, everyone reads a neighbour's after the barrier
    for (int i = threadIdx.x; i < payload; i += blockDim.x) data[(size_t)id * payload + i] = it + i;
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      atomicAdd(ctr, 1);
      const int target = (it + 1) * W;
      while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
      __threadfence();
    }
    __syncthreads();
    const int nb = (id + 1) % W;
    for (int i = threadIdx.x; i < payload; i += blockDim.x) acc += data[(size_t)nb * payload + i];
  }
  if (acc == -1.0f) data[0] = acc;
}

int main(int argc, char** argv) {
  int* ctr; float* data;
  hipMalloc(&ctr, 4); hipMalloc(&data, 64 << 20);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 200;
  int Ws[] = {8, 21, 32, 64, 128};
  for (int stride : {1, 8}) for (int payload : {0, 4096, 15360}) for (int W : Ws) {
    hipMemset(ctr, 0, 4);
    bar_kernel<<<W * stride, 256>>>(ctr, data, W, 2, stride, payload);
    hipDeviceSynchronize();
    hipMemset(ctr, 0, 4);
    hipEventRecord(e0);
    bar_kernel<<<W * stride, 256>>>(ctr, data, W, iters, stride, payload);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("stride %d payload %6d floats  W %3d : %.2f us per barrier+payload\n", stride, payload, W, ms * 1e3 / iters);
  }
  return 0;
}
