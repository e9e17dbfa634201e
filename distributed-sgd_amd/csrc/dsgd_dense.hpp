// Device code of libdsgd_hip, part 3 (gfx950 only): K8, the DENSE logistic mini-batch step of BASELINE.json
// configs[4] ("Synthetic dense 10M x 4096 logistic, mini-batch GEMV").  Included by dsgd_hip.hip.
//
// NO REFERENCE COUNTERPART: zifeo/distributed-sgd has exactly one model, the sparse hinge "SVM"
// (core/ml/SparseSVM.scala:11; Main.scala:67 "could use another model").  This kernel family is the optional variant
// north_star names; its oracle (oracle/dense_ref.py) restates textbook logistic regression in fp64 and is pinned by
// finite differences, not by anything in the reference: PARITY UNPINNED by construction.
//
//   z = X w,  p = sigmoid(z),  loss = mean(softplus(z) - y z),  g = X^T (p - y) / B,  w <- w - lr * g
//
// Roofline: one fp32 row of D = 4096 features is 16,388 bytes for 4 D flops: 1 flop per byte, far
// left of the ridge -- HBM-bound.  The row block is read ONCE: it stays in registers between the forward product and
// the gradient product.
//
// Why not MFMA (measured reasoning, MI355X_MICROARCH.md "Matrix cores"): with fp32 INPUTS the matrix pipe runs at
// the fp32 VECTOR rate (v_mfma_f32_32x32x2_f32: 64 flop/clk/SIMD), and a matrix-VECTOR product fills one column of
// the N dimension: 1/32 (32x32x2) or 1/16 (16x16x4) of every instruction's multiply-adds are useful.  At 1/16 the
// two products of a step would need ~45 % of the matrix pipe's cycles to keep up with HBM and the operand layout
// (rows across lanes) forces 64-byte load segments; v_fma on a lane-owns-columns layout needs 5 % of the VALU and
// loads 1 KB per wave instruction.  The matrix cores have nothing to offer a single fp32 GEMV.
#pragma once

constexpr int DN_ROWS = 8;          // rows per block and workgroup iteration
constexpr int DN_COLS = 8;          // columns per lane: two float4 (D / 2 apart)

struct DenseArgs {
  const float* __restrict__ X;      // n_rows x D, row-major
  const float* __restrict__ y;      // n_rows labels in {0, 1}
  const float* __restrict__ w;      // D
  float* __restrict__ gpart;        // gridDim.x x D per-workgroup sums of (p - y) * x  (null: evaluation only)
  double* __restrict__ lpart;       // gridDim.x x 2: sum of losses, correct predictions
  long long row_begin, row_end;
  int D;
};

// One workgroup of D / 8 lanes per row block: lane t owns columns [4t, 4t+4) and [D/2 + 4t, D/2 + 4t + 4) of every
// row (each wave-level load is 1 KB of one row).  Two workgroups share a CU (<= 128 VGPRs each): one computes while
// the other's loads are in flight.
__global__ void __launch_bounds__(1024) dsgd_dense_step_kernel(DenseArgs a) {
  __shared__ float zpart[16][DN_ROWS];
  __shared__ float rl[DN_ROWS];
  __shared__ double lred[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n_waves = blockDim.x >> 6;
  const int c0 = 4 * tid, c1 = a.D / 2 + 4 * tid;
  const float4 w0 = *reinterpret_cast<const float4*>(a.w + c0), w1 = *reinterpret_cast<const float4*>(a.w + c1);
  float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = g0;
  double loss_acc = 0.0, corr_acc = 0.0;   // (thread r < DN_ROWS accumulates row r of every block)
  const long long n_blocks = (a.row_end - a.row_begin + DN_ROWS - 1) / DN_ROWS;
  for (long long blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
    const long long r0 = a.row_begin + blk * DN_ROWS;
    float4 x0[DN_ROWS], x1[DN_ROWS];
#pragma unroll
    for (int i = 0; i < DN_ROWS; ++i) {
      const bool in = r0 + i < a.row_end;
      const float* row = a.X + (in ? r0 + i : a.row_begin) * (long long)a.D;
      x0[i] = *reinterpret_cast<const float4*>(row + c0);
      x1[i] = *reinterpret_cast<const float4*>(row + c1);
    }
    // forward: partial z per lane, wave sums, workgroup sums
#pragma unroll
    for (int i = 0; i < DN_ROWS; ++i) {
      float p = x0[i].x * w0.x;
      p = fmaf(x0[i].y, w0.y, p);
      p = fmaf(x0[i].z, w0.z, p);
      p = fmaf(x0[i].w, w0.w, p);
      p = fmaf(x1[i].x, w1.x, p);
      p = fmaf(x1[i].y, w1.y, p);
      p = fmaf(x1[i].z, w1.z, p);
      p = fmaf(x1[i].w, w1.w, p);
      p = group_sum<64>(p);
      if (lane == 0) zpart[wave][i] = p;
    }
    __syncthreads();
    if (tid < DN_ROWS) {
      float z = 0.0f;
      for (int wv = 0; wv < n_waves; ++wv) z += zpart[wv][tid];
      const bool in = r0 + tid < a.row_end;
      const float yy = in ? a.y[r0 + tid] : 0.0f;
      const float p = 1.0f / (1.0f + __expf(-z));
      rl[tid] = in ? p - yy : 0.0f;
      if (in) {
        // softplus(z) - y z, evaluated without overflow: max(z, 0) + log1p(exp(-|z|)) - y z
        loss_acc += (double)(fmaxf(z, 0.0f) + log1pf(__expf(-fabsf(z))) - yy * z);
        corr_acc += ((z > 0.0f) == (yy > 0.5f)) ? 1.0 : 0.0;
      }
    }
    __syncthreads();
    // gradient: the same registers
    if (a.gpart) {
#pragma unroll
      for (int i = 0; i < DN_ROWS; ++i) {
        const float r = rl[i];
        g0.x = fmaf(r, x0[i].x, g0.x);
        g0.y = fmaf(r, x0[i].y, g0.y);
        g0.z = fmaf(r, x0[i].z, g0.z);
        g0.w = fmaf(r, x0[i].w, g0.w);
        g1.x = fmaf(r, x1[i].x, g1.x);
        g1.y = fmaf(r, x1[i].y, g1.y);
        g1.z = fmaf(r, x1[i].z, g1.z);
        g1.w = fmaf(r, x1[i].w, g1.w);
      }
    }
  }
  if (a.gpart) {
    float* mine = a.gpart + (long long)blockIdx.x * a.D;
    *reinterpret_cast<float4*>(mine + c0) = g0;
    *reinterpret_cast<float4*>(mine + c1) = g1;
  }
  if (tid == 0) lred[0] = lred[1] = 0.0;
  __syncthreads();
  if (tid < DN_ROWS) {
    atomicAdd(&lred[0], loss_acc);
    atomicAdd(&lred[1], corr_acc);
  }
  __syncthreads();
  if (tid == 0) {
    a.lpart[2 * blockIdx.x] = lred[0];
    a.lpart[2 * blockIdx.x + 1] = lred[1];
  }
}

// ---- the MFMA variant north_star names ("mini-batch GEMV via MFMA"), selectable with DSGD_DENSE_MFMA=1 -----------
// Forward product on the matrix cores: v_mfma_f32_16x16x4_f32 with A = a 16-row x 4-column tile of X (lane l: row
// l % 16, column group l / 16) and B = the four weights of those columns broadcast over the N dimension, so every
// column of D holds the same partial z and lane (j, q) register r carries z of row 4q + r.  A matrix-VECTOR product
// uses 1/16 of the instruction's multiply-adds and the fp32 matrix rate equals the vector rate, so this cannot beat
// v_fma (header comment): the variant exists to be MEASURED next to it (bench.py dense_logistic.mfma_variant).
// One workgroup of D / 4 lanes per 16-row block: wave v owns columns [256v, 256v + 256); lane (i, q) loads, for each
// of 16 macro-steps m, the float4 X[row i][256v + 16m + 4q ..]: 64 VGPRs hold the tile for both products.  The
// gradient product stays on the VALU: r_i * x summed over the 16 rows with the DPP butterfly of group_sum<16>, lane i
// keeping the four sums of macro-step m == i.
// (Round 6 measured the tile turned round as well -- rows on l / 16, X as the B operand: g = X^T r on the matrix cores with
//  A[m][k] = r[row k] for every m, z = X w on the VALU with one 16-lane DPP sum per row.  An MFMA contracts over l / 16 only,
//  so ONE register layout serves one of the two products; and the 16 identical result rows of a matrix-VECTOR product cost
//  64 accumulator registers per lane, which leaves a 1,024-lane workgroup room for blocks of FOUR rows, one workgroup per
//  CU, two barriers per block: 243 us per 65,536-row step against 185 for the VALU kernel (0.55 / 0.73 of 8 TB/s by wall
//  time, tools/dense_check.py; 128 VGPRs + 56 bytes of scratch) -- slower than this form (0.61).  Not kept;
//  profiles/r06_dense_pmc_summary.txt.)
typedef float dn_f32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(1024) dsgd_dense_step_mfma_kernel(DenseArgs a) {
  extern __shared__ __attribute__((aligned(16))) float wl[];   // D weights
  __shared__ float zpart[16][16];
  __shared__ float rl[16];
  __shared__ double lred[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n_waves = blockDim.x >> 6;
  const int i = lane & 15, q = lane >> 4;
  for (int j = tid * 4; j < a.D; j += blockDim.x * 4) *reinterpret_cast<float4*>(wl + j) = *reinterpret_cast<const float4*>(a.w + j);
  __syncthreads();
  const int cw = 256 * wave;
  float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);   // columns cw + 16 i + 4 q .. + 3
  double loss_acc = 0.0, corr_acc = 0.0;
  const long long n_blocks = (a.row_end - a.row_begin + 15) / 16;
  for (long long blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
    const long long r0 = a.row_begin + blk * 16;
    const bool in_i = r0 + i < a.row_end;
    const float* row = a.X + (in_i ? r0 + i : a.row_begin) * (long long)a.D + cw + 4 * q;
    float4 x[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) x[m] = *reinterpret_cast<const float4*>(row + 16 * m);
    dn_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const float4 wv = *reinterpret_cast<const float4*>(wl + cw + 16 * m + 4 * q);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x[m].x, wv.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x[m].y, wv.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x[m].z, wv.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x[m].w, wv.w, acc, 0, 0, 0);
    }
    if (i == 0) {   // D[4q + r][j] is the same for every j: one lane per column group writes its four rows
      zpart[wave][4 * q + 0] = acc[0];
      zpart[wave][4 * q + 1] = acc[1];
      zpart[wave][4 * q + 2] = acc[2];
      zpart[wave][4 * q + 3] = acc[3];
    }
    __syncthreads();
    if (tid < 16) {
      float z = 0.0f;
      for (int wv = 0; wv < n_waves; ++wv) z += zpart[wv][tid];
      const bool in = r0 + tid < a.row_end;
      const float yy = in ? a.y[r0 + tid] : 0.0f;
      const float p = 1.0f / (1.0f + __expf(-z));
      rl[tid] = in ? p - yy : 0.0f;
      if (in) {
        loss_acc += (double)(fmaxf(z, 0.0f) + log1pf(__expf(-fabsf(z))) - yy * z);
        corr_acc += ((z > 0.0f) == (yy > 0.5f)) ? 1.0 : 0.0;
      }
    }
    __syncthreads();
    if (a.gpart) {
      const float r = rl[i];
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        const float sx = group_sum<16>(r * x[m].x), sy = group_sum<16>(r * x[m].y);
        const float sz = group_sum<16>(r * x[m].z), sw = group_sum<16>(r * x[m].w);
        if (m == i) {
          g4.x += sx;
          g4.y += sy;
          g4.z += sz;
          g4.w += sw;
        }
      }
    }
  }
  if (a.gpart) *reinterpret_cast<float4*>(a.gpart + (long long)blockIdx.x * a.D + cw + 16 * i + 4 * q) = g4;
  if (tid == 0) lred[0] = lred[1] = 0.0;
  __syncthreads();
  if (tid < 16) {
    atomicAdd(&lred[0], loss_acc);
    atomicAdd(&lred[1], corr_acc);
  }
  __syncthreads();
  if (tid == 0) {
    a.lpart[2 * blockIdx.x] = lred[0];
    a.lpart[2 * blockIdx.x + 1] = lred[1];
  }
}

// g = sum of the per-workgroup partials (fixed order: reproducible) and, without a communicator, the update in the
// same pass.  Block = 64 columns x 16 phases over the partials (one thread per column summing 512 partials
// one after the other took ~100 us -- more than the step kernel itself at batch 4,096).
__global__ void __launch_bounds__(1024) dsgd_dense_reduce_kernel(const float* __restrict__ gpart, int n_part, int D,
                                                                float* __restrict__ g, float* __restrict__ w,
                                                                float scale /* lr / batch; w == nullptr: no update */) {
  __shared__ float red[16][64];
  const int cx = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + cx;
  float s = 0.0f;
  if (j < D)
    for (int b = ph; b < n_part; b += 16) s += gpart[(long long)b * D + j];
  red[ph][cx] = s;
  __syncthreads();
  if (ph == 0 && j < D) {
    float t = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][cx];
    g[j] = t;
    if (w) w[j] -= scale * t;
  }
}
__global__ void __launch_bounds__(256) dsgd_dense_apply_kernel(float* __restrict__ w, const float* __restrict__ g, int D,
                                                              float scale /* lr / global batch */) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < D) w[j] -= scale * g[j];
}

// synthetic data on the device (configs[4]: "X generated on-device per shard (seed = rank)"): x ~ N(0, 1) / sqrt(D)
// from a counter-based generator (splitmix64 of the element index), y = [x . w* + 0.1 N(0,1) > 0] with planted
// w* ~ N(0, 1) shared by all shards.
__device__ __forceinline__ float dn_normal(unsigned long long key) {
  const unsigned long long a = hog_mix(key), b = hog_mix(key ^ 0x5851F42D4C957F2Dull);
  const float u1 = ((float)(a >> 40) + 1.0f) * (1.0f / 16777217.0f);   // (0, 1)
  const float u2 = (float)(b >> 40) * (1.0f / 16777216.0f);
  return sqrtf(-2.0f * __logf(u1)) * __cosf(6.2831853f * u2);
}
__global__ void __launch_bounds__(256) dsgd_dense_generate_kernel(float* __restrict__ X, float* __restrict__ y,
                                                                 long long n_rows, int D, unsigned long long seed) {
  __shared__ float red[4];
  const float inv = rsqrtf((float)D);
  for (long long row = blockIdx.x; row < n_rows; row += gridDim.x) {
    float dot = 0.0f;
    for (int j = threadIdx.x; j < D; j += blockDim.x) {
      const float x = dn_normal(seed * 0x9E3779B97F4A7C15ull + (unsigned long long)row * (unsigned long long)D + j) * inv;
      // planted separator: ONE problem for every shard (the seed -- the rank -- varies x and the label noise only;
      // ranks that all-reduce gradients of different separators would be training nothing)
      const float ws = dn_normal(0xD1B54A32D192ED03ull + j);
      X[row * (long long)D + j] = x;
      dot = fmaf(x, ws, dot);
    }
    dot = group_sum<64>(dot);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dot;
    __syncthreads();
    if (threadIdx.x == 0) {
      const float z = red[0] + red[1] + red[2] + red[3] + 0.1f * dn_normal(~seed * 0xA0761D6478BD642Full + (unsigned long long)row);
      y[row] = z > 0.0f ? 1.0f : 0.0f;
    }
  }
}
