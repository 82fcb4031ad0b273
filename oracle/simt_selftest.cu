// simt_selftest.cu — kernels of OUR OWN, written in the CUDA subset that oracle/simt_emul.h emulates, whose results
// are known in closed form.  tests/test_refemul_cpu.py runs them to check the emulator itself (shuffle semantics,
// votes, barriers, atomics, the texture unit, the launch rewrite) before trusting what the reference's kernels compute
// on it.  TEST INFRASTRUCTURE ONLY; built by oracle/Makefile into oracle/libsimt_selftest.so with the same
// `<<< >>>` rewrite and flags as the reference translation units (oracle/build_ref.sh).

__device__ unsigned int st_counter[4];

// out[0..5][tid]: shfl_down(3), shfl_up(2), shfl(idx 5), shfl_down(1, width 8), any(tid == 77), ballot(tid odd)
__global__ void StWarp(int *out, int n)
{
  const int tid = blockIdx.x*blockDim.x + threadIdx.x;
  int v = 1000 + tid;
  int a = __shfl_down_sync(0xffffffff, v, 3);
  int b = __shfl_up_sync(0xffffffff, v, 2);
  int c = __shfl_sync(0xffffffff, v, 5);
  int d = __shfl_down_sync(0xffffffff, v, 1, 8);
  int e = __any_sync(0xffffffff, tid==77);
  unsigned f = __ballot_sync(0xffffffff, tid&1);
  out[0*n + tid] = a;
  out[1*n + tid] = b;
  out[2*n + tid] = c;
  out[3*n + tid] = d;
  out[4*n + tid] = e;
  out[5*n + tid] = (int)f;
}

// block-wide inclusive scan through shared memory (log-step, two barriers per step) + a 2-D thread index check
__global__ void StScan(const int *in, int *out, int *tids)
{
  __shared__ int buf[2][256];
  const int t = threadIdx.y*blockDim.x + threadIdx.x;
  const int n = blockDim.x*blockDim.y;
  buf[0][t] = in[blockIdx.x*n + t];
  __syncthreads();
  int cur = 0;
  for (int d=1;d<n;d<<=1) {
    buf[1-cur][t] = buf[cur][t] + (t>=d ? buf[cur][t-d] : 0);
    cur = 1 - cur;
    __syncthreads();
  }
  out[blockIdx.x*n + t] = buf[cur][t];
  tids[blockIdx.x*n + t] = threadIdx.x + 100*threadIdx.y + 10000*blockIdx.x;
}

// atomics across blocks: counter[0] += 1 per thread, counter[1] = max(tid), counter[2] wraps at 9 (atomicInc),
// fsum accumulates 0.5 per thread (exact in fp32 for the sizes used)
__global__ void StAtomics(float *fsum)
{
  const unsigned tid = blockIdx.x*blockDim.x + threadIdx.x;
  atomicAdd(&st_counter[0], 1u);
  atomicMax(&st_counter[1], tid);
  atomicInc(&st_counter[2], 9);
  atomicAdd(fsum, 0.5f);
}

// early exit of some lanes before a barrier / a shuffle of the rest of the block (exited threads do not count)
__global__ void StExit(int *out)
{
  const int t = threadIdx.x;
  if (t>=32 && t<48)
    return;
  __shared__ int s[64];
  s[t] = t;
  __syncthreads();
  int v = s[t ^ 1];
  if (t>=48)
    v += __shfl_down_sync(0xffffffff, v, 4);
  out[t] = v;
}

__global__ void StTex(cudaTextureObject_t tex, const float *xy, float *out, int n)
{
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  if (i<n)
    out[i] = tex2D<float>(tex, xy[2*i+0], xy[2*i+1]);
}

extern "C" void st_warp(int *out, int nblocks, int nthreads)
{
  StWarp<<<nblocks, nthreads>>>(out, nblocks*nthreads);
}

extern "C" void st_scan(const int *in, int *out, int *tids, int nblocks, int bx, int by)
{
  dim3 threads(bx, by);
  StScan<<<nblocks, threads>>>(in, out, tids);
}

extern "C" void st_atomics(int nblocks, int nthreads, unsigned int *counters, float *fsum)
{
  float *d_sum;
  cudaMalloc((void **)&d_sum, sizeof(float));
  cudaMemset(d_sum, 0, sizeof(float));
  cudaMemset(st_counter, 0, sizeof(st_counter));
  StAtomics<<<nblocks, nthreads>>>(d_sum);
  cudaMemcpy(fsum, d_sum, sizeof(float), cudaMemcpyDeviceToHost);
  memcpy(counters, st_counter, sizeof(st_counter));
  cudaFree(d_sum);
}

extern "C" void st_exit(int *out)
{
  StExit<<<1, 64>>>(out);
}

extern "C" void st_tex(const float *img, int w, int h, int pitch, const float *xy, float *out, int n)
{
  struct cudaResourceDesc resDesc;
  memset(&resDesc, 0, sizeof(resDesc));
  resDesc.resType = cudaResourceTypePitch2D;
  resDesc.res.pitch2D.devPtr = (void *)img;
  resDesc.res.pitch2D.width = w;
  resDesc.res.pitch2D.height = h;
  resDesc.res.pitch2D.pitchInBytes = pitch*sizeof(float);
  resDesc.res.pitch2D.desc = cudaCreateChannelDesc<float>();
  struct cudaTextureDesc texDesc;
  memset(&texDesc, 0, sizeof(texDesc));
  texDesc.addressMode[0] = cudaAddressModeClamp;
  texDesc.addressMode[1] = cudaAddressModeClamp;
  texDesc.filterMode = cudaFilterModeLinear;
  texDesc.readMode = cudaReadModeElementType;
  texDesc.normalizedCoords = 0;
  cudaTextureObject_t tex = 0;
  cudaCreateTextureObject(&tex, &resDesc, &texDesc, NULL);
  StTex<<<(n + 63)/64, 64>>>(tex, xy, out, n);
  cudaDestroyTextureObject(tex);
}

extern "C" float st_fmul_rz(float a, float b) { return __fmul_rz(a, b); }
extern "C" int st_f2i(float a) { return simt_f2i_rz(a); }
