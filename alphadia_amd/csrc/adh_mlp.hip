// adh_mlp.hip - the target/decoy classifier of the FDR stage, trained and evaluated on the device.
// Network and training loop: FeedForwardNN (fdr/classifiers.py:497-532) and
// BinaryClassifierLegacyNewBatching.fit / predict_proba (fdr/classifiers.py:316-495):
//   BatchNorm1d(d) -> [Linear -> ReLU -> Dropout] x hidden -> Linear -> Softmax, BCELoss, Adam.
// The whole model is 11 k parameters: every workgroup keeps a padded copy in LDS and pushes a 16-row
// tile of the batch through forward and backward without touching HBM in between.  Every product
// of the tile (X W^T forward, delta W and delta^T A backward) is a chain of v_mfma_f32_16x16x4_f32
// instructions on 16 x 16 output blocks, operands read straight from LDS in the MFMA lane layout
// (row strides are 4 * odd floats, so the 16 x 4 operand reads are bank-conflict free).
// A fit call = adh_mlp_bn_kernel once (batch statistics of every distinct batch: they depend on the
// rows only, not on the parameters), then per step adh_mlp_train_kernel (one tile per workgroup,
// per-tile gradients) -> adh_mlp_adam_kernel (ordered reduction over the tiles + Adam).  Launches
// are queued back to back on the handle's stream; the host never waits inside a fit.  Results are
// deterministic for a given seed.  Included by adh_api.hip.

#define ADH_MLP_TR 16
#define ADH_MLP_THREADS 512
#define ADH_MLP_WAVES (ADH_MLP_THREADS / 64)
#define ADH_MLP_BN_ROWS 64

struct MlpArch {
    int n_linear;
    int dims[ADH_MLP_MAX_LINEAR + 1];
    int w_off[ADH_MLP_MAX_LINEAR], b_off[ADH_MLP_MAX_LINEAR];  // into the parameter vector
    int w_lds[ADH_MLP_MAX_LINEAR], b_lds[ADH_MLP_MAX_LINEAR];  // into the LDS copy (zero padded)
    int in_s[ADH_MLP_MAX_LINEAR];                              // LDS row stride of W_l (4 * odd)
    int a_off[ADH_MLP_MAX_LINEAR + 1];                          // layer columns inside a tile row
    int a_stride;                                               // floats per tile row (4 * odd)
    int x_stride;                                               // floats per row of the normalised input
    int n_params;
    int lds_params;                                             // floats of the LDS parameter copy
    float bn_eps;
};

namespace mlp {

typedef float floatx4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float uniform01(uint64_t seed, uint32_t step, uint32_t row, uint32_t layer, uint32_t j) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * ((uint64_t)step + 1);
    z ^= ((uint64_t)row << 32) | ((uint64_t)layer << 24) | j;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)(z >> 40) * (1.0f / 16777216.0f);
}

struct Lds {
    float *w, *act, *xh, *mean, *rstd, *y, *dl, *g;
};

__device__ __forceinline__ Lds carve(float *lds, const MlpArch &A) {
    Lds L;
    L.w = lds;  // [gamma d][beta d] then per layer W[out16][in_s], b[out16]
    L.act = L.w + A.lds_params;
    L.xh = L.act + ADH_MLP_TR * A.a_stride;
    L.mean = L.xh + ADH_MLP_TR * A.x_stride;
    L.rstd = L.mean + A.dims[0];
    L.y = L.rstd + A.dims[0];
    L.dl = L.y + ADH_MLP_TR;             // d loss / d (layer outputs) of the tile, laid out like act (training only)
    L.g = L.dl + ADH_MLP_TR * A.a_stride;  // per-tile gradient of every parameter (training only)
    return L;
}

// padded parameter image (maintained by the Adam kernel) -> LDS, straight 16-byte copies
__device__ __forceinline__ void load_params(const MlpArch &A, const Lds &L, const float *__restrict__ img) {
    const float4 *src = reinterpret_cast<const float4 *>(img);
    float4 *dst = reinterpret_cast<float4 *>(L.w);
    const int n4 = A.lds_params / 4;
    for (int base = threadIdx.x; base < n4; base += 10 * ADH_MLP_THREADS) {
        float4 t[10];  // all loads of a lane in flight before the first LDS write (one round for 80 KB)
#pragma unroll
        for (int u = 0; u < 10; ++u)
            if (base + u * ADH_MLP_THREADS < n4) t[u] = src[base + u * ADH_MLP_THREADS];
#pragma unroll
        for (int u = 0; u < 10; ++u)
            if (base + u * ADH_MLP_THREADS < n4) dst[base + u * ADH_MLP_THREADS] = t[u];
    }
}

// one 16 x 16 output block: acc += sum_k A[.][k] B[k][.] over kp (a multiple of 16) columns; the two
// operand streams advance by a_step / b_step floats per k.  Eight LDS reads are issued ahead of four
// MFMAs on two independent accumulators.
__device__ __forceinline__ floatx4 mfma_chain(const float *__restrict__ a, int a_step, const float *__restrict__ b,
                                              int b_step, int kp) {
    floatx4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
    float a0 = a[0], a1 = a[4 * a_step], a2 = a[8 * a_step], a3 = a[12 * a_step];
    float b0 = b[0], b1 = b[4 * b_step], b2 = b[8 * b_step], b3 = b[12 * b_step];
    for (int k0 = 16; k0 < kp; k0 += 16) {  // the next operands are in flight while these multiply
        const float na0 = a[(k0 + 0) * a_step], na1 = a[(k0 + 4) * a_step], na2 = a[(k0 + 8) * a_step],
                    na3 = a[(k0 + 12) * a_step];
        const float nb0 = b[(k0 + 0) * b_step], nb1 = b[(k0 + 4) * b_step], nb2 = b[(k0 + 8) * b_step],
                    nb3 = b[(k0 + 12) * b_step];
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b2, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, b3, acc1, 0, 0, 0);
        a0 = na0, a1 = na1, a2 = na2, a3 = na3;
        b0 = nb0, b1 = nb1, b2 = nb2, b3 = nb3;
    }
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b2, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, b3, acc1, 0, 0, 0);
    return acc0 + acc1;
}

// Linear layers of one tile: out[16 x N] = act[16 x K] W^T + b, one 16 x 16 block per wave and
// round.  MFMA operands: lane (i = lane % 16, q = lane / 16) supplies A[i][k0 + q] = act[row i]
// and B[k0 + q][i] = W[j0 + i]; D[4 q + rr][i] comes back in acc[rr].  TRAIN adds dropout
// (classifiers.py:520-524).
template <bool TRAIN>
__device__ __forceinline__ void forward_layers(const MlpArch &A, const Lds &L, float drop_p, float keep_scale,
                                               uint64_t seed, uint32_t step, uint32_t row0) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 15, q = lane >> 4;
    const int S = A.a_stride;
    for (int l = 0; l < A.n_linear; ++l) {
        const int in = A.dims[l], out = A.dims[l + 1], in_s = A.in_s[l];
        const int kp = (in + 15) & ~15, n_blocks = (out + 15) >> 4;
        const float *W = L.w + A.w_lds[l], *b = L.w + A.b_lds[l];
        const float *ap = L.act + i * S + A.a_off[l] + q;
        float *an = L.act + A.a_off[l + 1];
        const bool hidden = l + 1 < A.n_linear;
        for (int blk = wave; blk < n_blocks; blk += ADH_MLP_WAVES) {
            const int j = blk * 16 + i;
            const float *wp = W + j * in_s + q;
            const floatx4 acc = mfma_chain(ap, 1, wp, 1, kp);
            const float bias = b[j];  // zero in the padding
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int r = 4 * q + rr;
                float v = acc[rr] + bias;
                if (hidden) {
                    v = v > 0.0f ? v : 0.0f;
                    if (TRAIN && drop_p > 0.0f && v > 0.0f)
                        v = uniform01(seed, step, row0 + r, l, j) < drop_p ? 0.0f : v * keep_scale;
                }
                an[r * S + j] = j < out ? v : 0.0f;
            }
        }
        __syncthreads();
    }
}

// nn.Softmax(dim=1) of the last layer, in place; one thread per row
__device__ __forceinline__ void softmax_row(const MlpArch &A, float *z) {
    const int out = A.dims[A.n_linear];
    float m = z[0];
    for (int c = 1; c < out; ++c) m = fmaxf(m, z[c]);
    float s = 0.0f;
    for (int c = 0; c < out; ++c) {
        z[c] = expf(z[c] - m);
        s += z[c];
    }
    for (int c = 0; c < out; ++c) z[c] = z[c] / s;
}

}  // namespace mlp

// Batch statistics of BatchNorm1d in training mode for every distinct batch of a fit call:
// blockIdx.y = batch, blockIdx.x = chunk of ADH_MLP_BN_ROWS rows.  Every block sums its rows (shifted
// by the first row of the batch, in float64); the last block of a batch to finish adds the partial
// sums in chunk order and writes mean, biased variance and 1/sqrt(var + eps).
__global__ __launch_bounds__(256) void adh_mlp_bn_kernel(
    const float *__restrict__ X, int d, const int64_t *__restrict__ batch_start, int B, double *__restrict__ part, unsigned *__restrict__ ticket, double bn_eps,
    float *__restrict__ stats /* per batch: mean[d], var[d], rstd[d] */) {
    __shared__ double red[2][256];
    __shared__ bool last;
    const int tid = threadIdx.x;
    const int groups = 256 / 64;
    const int g = tid / 64, lane = tid % 64;
    const int batch = blockIdx.y, n_chunks = gridDim.x;
    const int64_t b0 = batch_start[batch];
    const int r0 = blockIdx.x * ADH_MLP_BN_ROWS, r1 = min(B, r0 + ADH_MLP_BN_ROWS);
    const int64_t pivot_row = b0;
    double *bpart = part + (int64_t)batch * n_chunks * 2 * d;
    for (int c0 = 0; c0 < d; c0 += 64) {
        const int c = c0 + lane;
        double s = 0.0, ss = 0.0;
        if (c < d) {
            const double pivot = (double)X[pivot_row * d + c];
#pragma unroll 8
            for (int r = r0 + g; r < r1; r += groups) {
                const double v = (double)X[(b0 + r) * d + c] - pivot;
                s += v;
                ss += v * v;
            }
        }
        red[0][tid] = s;
        red[1][tid] = ss;
        __syncthreads();
        if (g == 0 && c < d) {
            for (int qg = 1; qg < groups; ++qg) {
                s += red[0][qg * 64 + lane];
                ss += red[1][qg * 64 + lane];
            }
            bpart[((int64_t)blockIdx.x * 2 + 0) * d + c] = s;
            bpart[((int64_t)blockIdx.x * 2 + 1) * d + c] = ss;
        }
        __syncthreads();
    }
    __threadfence();
    if (tid == 0) last = atomicAdd(ticket + batch, 1u) == (unsigned)n_chunks - 1;
    __syncthreads();
    if (!last) return;
    __threadfence();
    float *bstats = stats + (int64_t)batch * 3 * d;
    for (int c = tid; c < d; c += 256) {
        double s = 0.0, ss = 0.0;
#pragma unroll 4
        for (int qc = 0; qc < n_chunks; ++qc) {
            s += bpart[((int64_t)qc * 2 + 0) * d + c];
            ss += bpart[((int64_t)qc * 2 + 1) * d + c];
        }
        const double ms = s / (double)B;
        double var = ss / (double)B - ms * ms;
        if (var < 0.0) var = 0.0;
        bstats[c] = (float)((double)X[pivot_row * d + c] + ms);
        bstats[d + c] = (float)var;
        bstats[2 * d + c] = (float)(1.0 / sqrt(var + bn_eps));
    }
    if (tid == 0) ticket[batch] = 0;
}

// x_train / y_train of a fit call as contiguous copies: batches become plain row ranges
__global__ void adh_mlp_take_rows_kernel(const float *__restrict__ X, const float *__restrict__ Y, int d,
                                         const int64_t *__restrict__ rows, int64_t n, float *__restrict__ Xt,
                                         float *__restrict__ Yt) {
    const int64_t total = n * d;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = idx / d;
        const int c = (int)(idx - r * d);
        Xt[idx] = X[rows[r] * d + c];
        if (c == 0) Yt[r] = Y[rows[r]];
    }
}

// Forward + backward of one 16-row tile; writes this tile's gradient of every parameter.
__global__ __launch_bounds__(ADH_MLP_THREADS) void adh_mlp_train_kernel(
    MlpArch A, const float *__restrict__ P, const float *__restrict__ X, const float *__restrict__ Y, int64_t b0,
    int B, const float *__restrict__ stats, float drop_p,
    float keep_scale, uint64_t seed, uint32_t step, float *__restrict__ gpart, float *__restrict__ loss_part) {
    extern __shared__ float lds_mlp[];
    const mlp::Lds L = mlp::carve(lds_mlp, A);
    const int tid = threadIdx.x, tile = blockIdx.x;
    const int wave = tid >> 6, lane = tid & 63, i = lane & 15, q = lane >> 4;
    const int d = A.dims[0], S = A.a_stride, XS = A.x_stride;
    const int row0 = tile * ADH_MLP_TR;
    const int rows_here = min(ADH_MLP_TR, B - row0);
    // the tile's rows are requested first, the parameter image streams in behind them
    const int dp = A.a_off[1];
    const bool x_in_regs = ADH_MLP_TR * dp <= 2 * ADH_MLP_THREADS;
    float xr[2] = {0.0f, 0.0f};
    if (x_in_regs) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int idx = tid + u * ADH_MLP_THREADS, r = idx / dp, c = idx % dp;
            if (r < rows_here && c < d) xr[u] = X[(b0 + row0 + r) * d + c];
        }
    }
    mlp::load_params(A, L, P);
    for (int c = tid; c < d; c += ADH_MLP_THREADS) {
        L.mean[c] = stats[c];
        L.rstd[c] = stats[2 * d + c];
    }
    if (tid < ADH_MLP_TR) L.y[tid] = tid < rows_here ? Y[b0 + row0 + tid] : 0.0f;
    __syncthreads();
    // BatchNorm1d, training mode (classifiers.py:519); the padding columns of the tile stay zero
    for (int idx = tid, u = 0; idx < ADH_MLP_TR * dp; idx += ADH_MLP_THREADS, ++u) {
        const int r = idx / dp, c = idx % dp;
        float xn = 0.0f, a0 = 0.0f;
        if (r < rows_here && c < d) {
            const float x = x_in_regs ? (u == 0 ? xr[0] : xr[1]) : X[(b0 + row0 + r) * d + c];
            xn = (x - L.mean[c]) * L.rstd[c];
            a0 = fmaf(xn, L.w[c], L.w[d + c]);
        }
        if (c < d) L.xh[r * XS + c] = xn;
        L.act[r * S + c] = a0;
    }
    __syncthreads();
    mlp::forward_layers<true>(A, L, drop_p, keep_scale, seed, step, (uint32_t)row0);

    // Softmax, nn.BCELoss (mean over B * out elements, log clamped at -100) and d loss / d logits
    const int out_dim = A.dims[A.n_linear];
    float *zl = L.act + A.a_off[A.n_linear];
    float *dzl = L.dl + A.a_off[A.n_linear];
    if (tid < 64) {
        float loss = 0.0f;
        if (tid < rows_here) {
            float *p = zl + tid * S;
            mlp::softmax_row(A, p);
            const float inv_n = 1.0f / ((float)B * (float)out_dim);
            const float y1 = L.y[tid];
            float g[ADH_MLP_MAX_LINEAR];  // out_dim <= 8
            float dot = 0.0f;
            for (int c = 0; c < out_dim; ++c) {
                const float t = c == 1 ? y1 : (c == 0 ? 1.0f - y1 : 0.0f);
                const float lp = fmaxf(logf(p[c]), -100.0f), lq = fmaxf(logf(1.0f - p[c]), -100.0f);
                loss -= t * lp + (1.0f - t) * lq;
                g[c] = (p[c] - t) / fmaxf((1.0f - p[c]) * p[c], 1e-12f) * inv_n;
                dot = fmaf(g[c], p[c], dot);
            }
            for (int c = 0; c < 16; ++c) dzl[tid * S + c] = c < out_dim ? p[c] * (g[c] - dot) : 0.0f;
        } else if (tid < ADH_MLP_TR) {
            for (int c = 0; c < 16; ++c) dzl[tid * S + c] = 0.0f;
        }
        for (int o = 32; o > 0; o >>= 1) loss += __shfl_down(loss, o, 64);
        if (tid == 0) loss_part[tile] = loss;
    }
    __syncthreads();
    for (int l = A.n_linear - 1; l >= 0; --l) {
        const int in = A.dims[l], out = A.dims[l + 1], in_s = A.in_s[l];
        const float *W = L.w + A.w_lds[l];
        const float *delta = L.dl + A.a_off[l + 1];  // d loss / d (outputs of layer l)
        const float *ap = L.act + A.a_off[l];        // inputs of layer l (read only from here on)
        float *dprev = L.dl + A.a_off[l];
        const int nb = (out + 15) >> 4, kb = (in + 15) >> 4, np16 = (out + 15) & ~15;
        // one phase per layer: the first kb blocks are d loss / d inputs, the others the weight gradient
        for (int blk = wave; blk < kb + nb * kb; blk += ADH_MLP_WAVES) {
            if (blk < kb) {
                // d a[r][k] = sum_n delta[r][n] W[n][k], times the ReLU/dropout mask of layer l - 1
                const int k = blk * 16 + i;
                const mlp::floatx4 acc = mlp::mfma_chain(delta + i * S + q, 1, W + q * in_s + k, in_s, np16);
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int r = 4 * q + rr;
                    float v = acc[rr];
                    if (l > 0) v = ap[r * S + k] > 0.0f ? v * keep_scale : 0.0f;
                    dprev[r * S + k] = k < in ? v : 0.0f;
                }
            } else {
                // dW[n][k] = sum_r delta[r][n] a[r][k]: A operand = delta^T, B operand = a, 4 MFMAs over the rows
                const int wb = blk - kb;
                const int n0 = (wb / kb) * 16, k0 = (wb % kb) * 16;
                mlp::floatx4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int r0 = 0; r0 < ADH_MLP_TR; r0 += 4)
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(delta[(r0 + q) * S + n0 + i], ap[(r0 + q) * S + k0 + i],
                                                               acc, 0, 0, 0);
                const int k = k0 + i;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int n = n0 + 4 * q + rr;
                    if (n < out && k < in) L.g[A.w_off[l] + n * in + k] = acc[rr];
                }
            }
        }
        for (int j = tid; j < out; j += ADH_MLP_THREADS) {
            float s = 0.0f;
#pragma unroll
            for (int r = 0; r < ADH_MLP_TR; ++r) s += delta[r * S + j];
            L.g[A.b_off[l] + j] = s;
        }
        __syncthreads();
    }
    // BatchNorm affine parameters
    for (int c = tid; c < d; c += ADH_MLP_THREADS) {
        float sg = 0.0f, sb = 0.0f;
#pragma unroll
        for (int r = 0; r < ADH_MLP_TR; ++r) {
            const float g = L.dl[r * S + c];
            sg = fmaf(g, L.xh[r * XS + c], sg);
            sb += g;
        }
        L.g[c] = sg;
        L.g[d + c] = sb;
    }
    __syncthreads();
    // one coalesced write of the tile's gradient
    float *gt = gpart + (int64_t)tile * A.n_params;
    for (int idx = tid; idx < A.n_params; idx += ADH_MLP_THREADS) gt[idx] = L.g[idx];
}

// Sum the tile gradients (8 tile groups per parameter in parallel, fixed order), then
// torch.optim.Adam (weight decay added to the gradient, classifiers.py:356-360); block 0 also
// updates the BatchNorm running statistics and the loss.
#define ADH_MLP_ADAM_PARAMS 32
__global__ __launch_bounds__(256) void adh_mlp_adam_kernel(
    int n_params, int n_tiles, const float *__restrict__ gpart, float *__restrict__ P /* padded image */,
    const int *__restrict__ pos /* parameter -> image position */, float *__restrict__ m,
    float *__restrict__ v, float weight_decay, float one_minus_beta1, float beta2, float one_minus_beta2,
    float step_size, float bc2_sqrt, float eps, int d, const float *__restrict__ stats, float *__restrict__ rm,
    float *__restrict__ rv, float momentum, float unbias, const float *__restrict__ loss_part, float loss_scale,
    float *__restrict__ loss_out) {
    __shared__ float red[8][ADH_MLP_ADAM_PARAMS];
    const int lp = threadIdx.x % ADH_MLP_ADAM_PARAMS, grp = threadIdx.x / ADH_MLP_ADAM_PARAMS;
    const int i = blockIdx.x * ADH_MLP_ADAM_PARAMS + lp;
    float g = 0.0f;
    if (i < n_params) {
        const int per = (n_tiles + 7) / 8;
        const int t1 = min(n_tiles, (grp + 1) * per);
#pragma unroll 8
        for (int t = grp * per; t < t1; ++t) g += gpart[(int64_t)t * n_params + i];
    }
    red[grp][lp] = g;
    __syncthreads();
    if (grp == 0 && i < n_params) {
#pragma unroll
        for (int qg = 1; qg < 8; ++qg) g += red[qg][lp];
        const int ip = pos[i];
        const float p = P[ip];
        g = fmaf(weight_decay, p, g);
        float mi = m[i], vi = v[i];
        mi = fmaf(g - mi, one_minus_beta1, mi);
        vi = fmaf(one_minus_beta2 * g, g, vi * beta2);
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        m[i] = mi;
        v[i] = vi;
        P[ip] = p - step_size * (mi / denom);
    }
    if (blockIdx.x == 0) {
        for (int c = threadIdx.x; c < d; c += 256) {
            rm[c] = fmaf(momentum, stats[c] - rm[c], rm[c]);
            rv[c] = fmaf(momentum, stats[d + c] * unbias - rv[c], rv[c]);
        }
        if (threadIdx.x < 64 && loss_out) {
            float s = 0.0f;
            for (int t = threadIdx.x; t < n_tiles; t += 64) s += loss_part[t];
            for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
            if (threadIdx.x == 0) *loss_out = s * loss_scale;
        }
    }
}

// network.eval() forward: running statistics, no dropout; tiles strided over the workgroups
__global__ __launch_bounds__(ADH_MLP_THREADS) void adh_mlp_predict_kernel(
    MlpArch A, const float *__restrict__ P, const float *__restrict__ rm, const float *__restrict__ rv,
    const float *__restrict__ X, const int64_t *__restrict__ rows, int64_t n, float *__restrict__ proba) {
    extern __shared__ float lds_mlp[];
    const mlp::Lds L = mlp::carve(lds_mlp, A);
    const int tid = threadIdx.x;
    const int d = A.dims[0], S = A.a_stride, out_dim = A.dims[A.n_linear], dp = A.a_off[1];
    mlp::load_params(A, L, P);
    for (int c = tid; c < d; c += ADH_MLP_THREADS) {
        L.mean[c] = rm[c];
        L.rstd[c] = (float)(1.0 / sqrt((double)rv[c] + (double)A.bn_eps));
    }
    __syncthreads();
    const int64_t n_tiles = (n + ADH_MLP_TR - 1) / ADH_MLP_TR;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t row0 = tile * ADH_MLP_TR;
        const int rows_here = (int)min((int64_t)ADH_MLP_TR, n - row0);
        for (int idx = tid; idx < ADH_MLP_TR * dp; idx += ADH_MLP_THREADS) {
            const int r = idx / dp, c = idx % dp;
            float a0 = 0.0f;
            if (r < rows_here && c < d) {
                const int64_t src = rows ? rows[row0 + r] : row0 + r;
                a0 = fmaf((X[src * d + c] - L.mean[c]) * L.rstd[c], L.w[c], L.w[d + c]);
            }
            L.act[r * S + c] = a0;
        }
        __syncthreads();
        mlp::forward_layers<false>(A, L, 0.0f, 1.0f, 0, 0, 0);
        if (tid < rows_here) {
            float *p = L.act + A.a_off[A.n_linear] + tid * S;
            mlp::softmax_row(A, p);
            for (int c = 0; c < out_dim; ++c) proba[(row0 + tid) * out_dim + c] = p[c];
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct adh_mlp {
    adh_handle *h = nullptr;
    MlpArch A{};
    float bn_momentum = 0.1f;
    float *d_P = nullptr, *d_m = nullptr, *d_v = nullptr, *d_rm = nullptr, *d_rv = nullptr, *d_stats = nullptr;
    int64_t nbt = 0;
    float *d_X = nullptr, *d_Y = nullptr;
    int64_t n_rows = 0;
    bool has_y = false;
    int64_t *d_rows = nullptr;
    int64_t rows_cap = 0;
    float *d_gpart = nullptr, *d_loss_part = nullptr, *d_loss = nullptr;
    int64_t tiles_cap = 0, loss_cap = 0;
    double *d_bn_part = nullptr;
    int64_t bn_cap = 0;
    unsigned *d_ticket = nullptr;
    int64_t *d_batch_start = nullptr;
    int64_t ticket_cap = 0, stats_cap = 0, batch_cap = 0;
    int *d_pos = nullptr;           // parameter -> position in the padded image (d_P)
    std::vector<int> pos;
    float *d_Xt = nullptr, *d_Yt = nullptr;  // x_train / y_train of the current fit call
    int64_t xt_cap = 0, yt_cap = 0;
    size_t lds_bytes = 0;
    double fit_ms = 0.0, predict_ms = 0.0;
    // rows staged straight from the scoring tables in HBM (adh_fdr_device.hip)
    int64_t *d_rowmap = nullptr;   // [n_rows] candidate row of every staged row
    uint8_t *d_decoy = nullptr;    // [n_rows]
    float *d_proba = nullptr;      // [n_rows][out] of adh_mlp_predict_resident
    int64_t n_table = 0;           // rows of the table the stage came from
    bool proba_ready = false;
};

namespace {

int mlp_layout(const adh_mlp_arch_t *a, MlpArch &A) {
    if (!a || a->n_linear < 1 || a->n_linear > ADH_MLP_MAX_LINEAR)
        return fail(ADH_ERR_INVALID_ARGUMENT, "n_linear must be 1.." + std::to_string(ADH_MLP_MAX_LINEAR));
    A = MlpArch{};
    A.n_linear = a->n_linear;
    auto odd4 = [](int n) {  // round up to 4 * odd: rows of 16 x 4 MFMA operands then hit distinct banks
        n = (n + 3) & ~3;
        return (n / 4) % 2 ? n : n + 4;
    };
    int col = 0;
    for (int l = 0; l <= a->n_linear; ++l) {
        if (a->dims[l] < 1 || a->dims[l] > 1024) return fail(ADH_ERR_INVALID_ARGUMENT, "layer sizes must be 1..1024");
        A.dims[l] = a->dims[l];
        A.a_off[l] = col;
        col += (a->dims[l] + 15) & ~15;
    }
    if (a->dims[a->n_linear] > ADH_MLP_MAX_LINEAR || a->dims[a->n_linear] < 2)
        return fail(ADH_ERR_UNSUPPORTED, "output_dim must be 2.." + std::to_string(ADH_MLP_MAX_LINEAR));
    A.a_stride = odd4(col);
    A.x_stride = a->dims[0] | 1;
    int off = 2 * a->dims[0], lds = 2 * a->dims[0];
    lds = (lds + 3) & ~3;
    for (int l = 0; l < a->n_linear; ++l) {
        const int in = a->dims[l], out = a->dims[l + 1], out16 = (out + 15) & ~15;
        A.w_off[l] = off;
        off += in * out;
        A.b_off[l] = off;
        off += out;
        A.in_s[l] = odd4((in + 15) & ~15);
        A.w_lds[l] = lds;
        lds += out16 * A.in_s[l];
        A.b_lds[l] = lds;
        lds += out16;
    }
    A.n_params = off;
    A.lds_params = (lds + 16 + 3) & ~3;  // slack: the last operand rows may be read 16 floats past their end
    A.bn_eps = a->bn_eps;
    return ADH_OK;
}

size_t mlp_lds_bytes(const MlpArch &A) {
    return sizeof(float) * ((size_t)A.lds_params + 2 * (size_t)ADH_MLP_TR * A.a_stride +
                            (size_t)ADH_MLP_TR * A.x_stride + 2 * (size_t)A.dims[0] + ADH_MLP_TR + (size_t)A.n_params);
}

template <typename T>
int mlp_reserve(T **p, int64_t *cap, int64_t want) {
    if (want <= *cap) return ADH_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    HIP_TRY(hipMalloc((void **)p, (size_t)want * sizeof(T)));
    *cap = want;
    return ADH_OK;
}

}  // namespace

int adh_mlp_param_count(const adh_mlp_arch_t *arch, int64_t *n_params) {
    MlpArch A;
    int rc = mlp_layout(arch, A);
    if (rc != ADH_OK) return rc;
    if (!n_params) return fail(ADH_ERR_INVALID_ARGUMENT, "NULL argument");
    *n_params = A.n_params;
    return ADH_OK;
}

int adh_mlp_destroy(adh_mlp_t *m) {
    if (!m) return ADH_OK;
    (void)hipSetDevice(m->h->device);
    void *ptrs[] = {m->d_P, m->d_m, m->d_v, m->d_rm, m->d_rv, m->d_stats, m->d_X, m->d_Y, m->d_rows, m->d_gpart,
                    m->d_loss_part, m->d_loss, m->d_bn_part, m->d_ticket, m->d_batch_start, m->d_pos, m->d_Xt, m->d_Yt,
                    m->d_rowmap, m->d_decoy, m->d_proba};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    delete m;
    return ADH_OK;
}

int adh_mlp_create(adh_handle_t *h, const adh_mlp_arch_t *arch, adh_mlp_t **out) {
    if (!h || !arch || !out) return fail(ADH_ERR_INVALID_ARGUMENT, "NULL argument");
    MlpArch A;
    int rc = mlp_layout(arch, A);
    if (rc != ADH_OK) return rc;
    const size_t lds = mlp_lds_bytes(A);
    if (lds > 160 * 1024) return fail(ADH_ERR_UNSUPPORTED, "network does not fit the 160 KB of LDS");
    HIP_TRY(hipSetDevice(h->device));
    adh_mlp *m = new adh_mlp();
    m->h = h;
    m->A = A;
    m->bn_momentum = arch->bn_momentum;
    m->lds_bytes = lds;
    const int d = A.dims[0];
    hipError_t e = hipSuccess;
    auto alloc = [&](void **p, size_t bytes) {
        if (e == hipSuccess) e = hipMalloc(p, bytes);
        if (e == hipSuccess) e = hipMemsetAsync(*p, 0, bytes, h->stream);
    };
    alloc((void **)&m->d_P, (size_t)A.lds_params * 4);
    alloc((void **)&m->d_pos, (size_t)A.n_params * 4);
    m->pos.resize((size_t)A.n_params);
    for (int c = 0; c < 2 * d; ++c) m->pos[(size_t)c] = c;
    for (int l = 0; l < A.n_linear; ++l) {
        const int in = A.dims[l], out = A.dims[l + 1];
        for (int j = 0; j < out; ++j) {
            for (int k = 0; k < in; ++k) m->pos[(size_t)(A.w_off[l] + j * in + k)] = A.w_lds[l] + j * A.in_s[l] + k;
            m->pos[(size_t)(A.b_off[l] + j)] = A.b_lds[l] + j;
        }
    }
    if (e == hipSuccess)
        e = hipMemcpyAsync(m->d_pos, m->pos.data(), (size_t)A.n_params * 4, hipMemcpyHostToDevice, h->stream);
    alloc((void **)&m->d_m, (size_t)A.n_params * 4);
    alloc((void **)&m->d_v, (size_t)A.n_params * 4);
    alloc((void **)&m->d_rm, (size_t)d * 4);
    alloc((void **)&m->d_rv, (size_t)d * 4);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) {
        adh_mlp_destroy(m);
        (void)hipGetLastError();
        return fail(e == hipErrorOutOfMemory ? ADH_ERR_OUT_OF_MEMORY : ADH_ERR_HIP,
                    std::string("adh_mlp_create: ") + hipGetErrorString(e));
    }
    (void)hipFuncSetAttribute((const void *)adh_mlp_train_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)adh_mlp_predict_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    (void)hipGetLastError();
    *out = m;
    return ADH_OK;
}

int adh_mlp_set_state(adh_mlp_t *m, const float *params, const float *running_mean, const float *running_var,
                      int64_t nbt) {
    if (!m || !params || !running_mean || !running_var) return fail(ADH_ERR_INVALID_ARGUMENT, "NULL argument");
    HIP_TRY(hipSetDevice(m->h->device));
    hipStream_t st = m->h->stream;
    const int d = m->A.dims[0];
    std::vector<float> img((size_t)m->A.lds_params, 0.0f);
    for (int i = 0; i < m->A.n_params; ++i) img[(size_t)m->pos[(size_t)i]] = params[i];
    HIP_TRY(hipMemcpyAsync(m->d_P, img.data(), img.size() * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(m->d_rm, running_mean, (size_t)d * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(m->d_rv, running_var, (size_t)d * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    m->nbt = nbt;
    return ADH_OK;
}

int adh_mlp_get_state(adh_mlp_t *m, float *params, float *running_mean, float *running_var, int64_t *nbt) {
    if (!m || !params || !running_mean || !running_var) return fail(ADH_ERR_INVALID_ARGUMENT, "NULL argument");
    HIP_TRY(hipSetDevice(m->h->device));
    hipStream_t st = m->h->stream;
    const int d = m->A.dims[0];
    std::vector<float> img((size_t)m->A.lds_params, 0.0f);
    HIP_TRY(hipMemcpyAsync(img.data(), m->d_P, img.size() * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(running_mean, m->d_rm, (size_t)d * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(running_var, m->d_rv, (size_t)d * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (int i = 0; i < m->A.n_params; ++i) params[i] = img[(size_t)m->pos[(size_t)i]];
    if (nbt) *nbt = m->nbt;
    return ADH_OK;
}

int adh_mlp_stage_rows(adh_mlp_t *m, const float *x, int64_t n, int32_t d, const float *y) {
    if (!m || n < 0 || (n > 0 && !x)) return fail(ADH_ERR_INVALID_ARGUMENT, "NULL argument");
    if (d != m->A.dims[0])
        return fail(ADH_ERR_INVALID_ARGUMENT, "feature matrix has " + std::to_string(d) + " columns, the network " +
                                                  std::to_string(m->A.dims[0]));
    HIP_TRY(hipSetDevice(m->h->device));
    hipStream_t st = m->h->stream;
    if (m->d_X) (void)hipFree(m->d_X);
    if (m->d_Y) (void)hipFree(m->d_Y);
    m->d_X = m->d_Y = nullptr;
    m->n_rows = 0;
    m->has_y = false;
    HIP_TRY(hipMalloc((void **)&m->d_X, std::max<size_t>((size_t)n * d * 4, 4)));
    HIP_TRY(hipMalloc((void **)&m->d_Y, std::max<size_t>((size_t)n * 4, 4)));
    if (n > 0) {
        HIP_TRY(hipMemcpyAsync(m->d_X, x, (size_t)n * d * 4, hipMemcpyHostToDevice, st));
        if (y) HIP_TRY(hipMemcpyAsync(m->d_Y, y, (size_t)n * 4, hipMemcpyHostToDevice, st));
    }
    HIP_TRY(hipStreamSynchronize(st));
    m->n_rows = n;
    m->has_y = y != nullptr;
    return ADH_OK;
}

int adh_mlp_fit(adh_mlp_t *m, const adh_mlp_fit_t *f, float *train_loss) {
    if (!m || !f) return fail(ADH_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!m->d_X || !m->has_y) return fail(ADH_ERR_NOT_STAGED, "stage the rows and their targets first (adh_mlp_stage_rows)");
    if (f->n_steps < 0 || f->n_train < 0 || f->batch_size < 2 || f->first_step < 0 ||
        (f->n_steps > 0 && (!f->train_rows || !f->batch_start)))
        return fail(ADH_ERR_INVALID_ARGUMENT, "invalid training schedule (batch_size must be at least 2)");
    if (!(f->dropout >= 0.0f && f->dropout < 1.0f)) return fail(ADH_ERR_INVALID_ARGUMENT, "dropout must be in [0, 1)");
    for (int64_t i = 0; i < f->n_train; ++i)
        if (f->train_rows[i] < 0 || f->train_rows[i] >= m->n_rows)
            return fail(ADH_ERR_INVALID_ARGUMENT, "train_rows out of range");
    for (int64_t s = 0; s < f->n_steps; ++s)
        if (f->batch_start[s] < 0 || f->batch_start[s] + f->batch_size > f->n_train)
            return fail(ADH_ERR_INVALID_ARGUMENT, "batch outside train_rows");
    m->fit_ms = 0.0;
    if (f->n_steps == 0) return ADH_OK;
    HIP_TRY(hipSetDevice(m->h->device));
    hipStream_t st = m->h->stream;
    const MlpArch &A = m->A;
    const int B = f->batch_size, d = A.dims[0];
    const int n_tiles = (B + ADH_MLP_TR - 1) / ADH_MLP_TR;
    const int n_chunks = (B + ADH_MLP_BN_ROWS - 1) / ADH_MLP_BN_ROWS;
    int rc;
    if ((rc = mlp_reserve(&m->d_rows, &m->rows_cap, f->n_train)) != ADH_OK) return rc;
    if (n_tiles > m->tiles_cap) {
        if (m->d_gpart) (void)hipFree(m->d_gpart);
        if (m->d_loss_part) (void)hipFree(m->d_loss_part);
        m->d_gpart = m->d_loss_part = nullptr;
        m->tiles_cap = 0;
        HIP_TRY(hipMalloc((void **)&m->d_gpart, (size_t)n_tiles * A.n_params * 4));
        HIP_TRY(hipMalloc((void **)&m->d_loss_part, (size_t)n_tiles * 4));
        m->tiles_cap = n_tiles;
    }
    if ((rc = mlp_reserve(&m->d_loss, &m->loss_cap, f->n_steps)) != ADH_OK) return rc;
    // distinct batches of this call: their statistics depend on the rows only
    std::vector<int64_t> uniq(f->batch_start, f->batch_start + f->n_steps);
    std::sort(uniq.begin(), uniq.end());
    uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
    const int64_t n_uniq = (int64_t)uniq.size();
    if ((rc = mlp_reserve(&m->d_bn_part, &m->bn_cap, n_uniq * n_chunks * 2 * d)) != ADH_OK) return rc;
    if ((rc = mlp_reserve(&m->d_stats, &m->stats_cap, n_uniq * 3 * d)) != ADH_OK) return rc;
    if ((rc = mlp_reserve(&m->d_batch_start, &m->batch_cap, n_uniq)) != ADH_OK) return rc;
    if (n_uniq > m->ticket_cap) {
        if ((rc = mlp_reserve(&m->d_ticket, &m->ticket_cap, n_uniq)) != ADH_OK) return rc;
        HIP_TRY(hipMemsetAsync(m->d_ticket, 0, (size_t)n_uniq * 4, st));
    }
    if ((rc = mlp_reserve(&m->d_Xt, &m->xt_cap, f->n_train * d)) != ADH_OK) return rc;
    if ((rc = mlp_reserve(&m->d_Yt, &m->yt_cap, f->n_train)) != ADH_OK) return rc;
    HIP_TRY(hipMemcpyAsync(m->d_rows, f->train_rows, (size_t)f->n_train * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(m->d_batch_start, uniq.data(), (size_t)n_uniq * 8, hipMemcpyHostToDevice, st));
    if (f->first_step == 0) {
        HIP_TRY(hipMemsetAsync(m->d_m, 0, (size_t)A.n_params * 4, st));
        HIP_TRY(hipMemsetAsync(m->d_v, 0, (size_t)A.n_params * 4, st));
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipEventRecord(e0, st));
    hipLaunchKernelGGL(adh_mlp_take_rows_kernel, dim3(2048), dim3(256), 0, st, m->d_X, m->d_Y, d, m->d_rows,
                       f->n_train, m->d_Xt, m->d_Yt);
    for (int64_t u0 = 0; u0 < n_uniq; u0 += 32768) {
        const int64_t nu = std::min<int64_t>(32768, n_uniq - u0);
        hipLaunchKernelGGL(adh_mlp_bn_kernel, dim3(n_chunks, (unsigned)nu), dim3(256), 0, st, m->d_Xt, d,
                           m->d_batch_start + u0, B, m->d_bn_part + u0 * n_chunks * 2 * d, m->d_ticket + u0,
                           (double)A.bn_eps, m->d_stats + u0 * 3 * d);
    }
    const float keep_scale = 1.0f / (1.0f - f->dropout);
    const float unbias = (float)((double)B / (double)(B - 1));
    const int adam_blocks = (A.n_params + ADH_MLP_ADAM_PARAMS - 1) / ADH_MLP_ADAM_PARAMS;
    for (int64_t s = 0; s < f->n_steps; ++s) {
        const int64_t t = f->first_step + s + 1;
        const double bc1 = 1.0 - pow((double)f->beta1, (double)t);
        const double bc2 = 1.0 - pow((double)f->beta2, (double)t);
        const int64_t slot = std::lower_bound(uniq.begin(), uniq.end(), f->batch_start[s]) - uniq.begin();
        const float *stats = m->d_stats + slot * 3 * d;
        hipLaunchKernelGGL(adh_mlp_train_kernel, dim3(n_tiles), dim3(ADH_MLP_THREADS), m->lds_bytes, st, A, m->d_P,
                           m->d_Xt, m->d_Yt, f->batch_start[s], B, stats, f->dropout, keep_scale, f->seed,
                           (uint32_t)(t - 1), m->d_gpart, m->d_loss_part);
        hipLaunchKernelGGL(adh_mlp_adam_kernel, dim3(adam_blocks), dim3(256), 0, st, A.n_params, n_tiles, m->d_gpart,
                           m->d_P, m->d_pos, m->d_m, m->d_v, f->weight_decay, 1.0f - f->beta1, f->beta2, 1.0f - f->beta2,
                           (float)((double)f->learning_rate / bc1), (float)sqrt(bc2), f->eps, d, stats, m->d_rm,
                           m->d_rv, m->bn_momentum, unbias, m->d_loss_part,
                           1.0f / ((float)B * (float)A.dims[A.n_linear]), m->d_loss + s);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(e1, st));
    if (train_loss) HIP_TRY(hipMemcpyAsync(train_loss, m->d_loss, (size_t)f->n_steps * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    float ms = 0.0f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    m->fit_ms = ms;
    m->nbt += f->n_steps;
    return ADH_OK;
}

int adh_mlp_predict(adh_mlp_t *m, const int64_t *rows, int64_t n, float *proba) {
    if (!m || n < 0 || (n > 0 && !proba)) return fail(ADH_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!m->d_X) return fail(ADH_ERR_NOT_STAGED, "stage the rows first (adh_mlp_stage_rows)");
    if (!rows && n != m->n_rows) return fail(ADH_ERR_INVALID_ARGUMENT, "n must be the number of staged rows when rows is NULL");
    if (rows)
        for (int64_t i = 0; i < n; ++i)
            if (rows[i] < 0 || rows[i] >= m->n_rows) return fail(ADH_ERR_INVALID_ARGUMENT, "rows out of range");
    m->predict_ms = 0.0;
    if (n == 0) return ADH_OK;
    HIP_TRY(hipSetDevice(m->h->device));
    hipStream_t st = m->h->stream;
    const int out_dim = m->A.dims[m->A.n_linear];
    int rc;
    if (rows) {
        if ((rc = mlp_reserve(&m->d_rows, &m->rows_cap, n)) != ADH_OK) return rc;
        HIP_TRY(hipMemcpyAsync(m->d_rows, rows, (size_t)n * 8, hipMemcpyHostToDevice, st));
    }
    float *d_out = nullptr;
    HIP_TRY(hipMalloc((void **)&d_out, (size_t)n * out_dim * 4));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, st);
    const int64_t n_tiles = (n + ADH_MLP_TR - 1) / ADH_MLP_TR;
    const unsigned grid = (unsigned)std::min<int64_t>(n_tiles, 256 * 8);
    hipLaunchKernelGGL(adh_mlp_predict_kernel, dim3(grid), dim3(ADH_MLP_THREADS), m->lds_bytes, st, m->A, m->d_P,
                       m->d_rm, m->d_rv, m->d_X, rows ? m->d_rows : nullptr, n, d_out);
    (void)hipEventRecord(e1, st);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(proba, d_out, (size_t)n * out_dim * 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    float ms = 0.0f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(d_out);
    if (e != hipSuccess) return fail(ADH_ERR_HIP, std::string("adh_mlp_predict: ") + hipGetErrorString(e));
    m->predict_ms = ms;
    m->h->d2h_bytes += (uint64_t)n * out_dim * 4;
    return ADH_OK;
}

int adh_mlp_time_ms(adh_mlp_t *m, double *fit_ms, double *predict_ms) {
    if (!m) return fail(ADH_ERR_INVALID_ARGUMENT, "NULL argument");
    if (fit_ms) *fit_ms = m->fit_ms;
    if (predict_ms) *predict_ms = m->predict_ms;
    return ADH_OK;
}
