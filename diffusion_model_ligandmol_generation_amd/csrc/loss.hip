// EDM noise / loss step of the trainer (train.py:267-452) fused into three kernels:
//   ligdiff_edm_prepare : sigma, scalings, noisy and preconditioned coordinates (train.py:286-312)
//   ligdiff_loss_partial: per-rank sums of every loss family (train.py:454-473, :475-498,
//                         :603-623, :625-663), one wave per molecule, fixed reduction order
//   ligdiff_loss_finish : means over the GLOBAL batch, the NaN/Inf skip rule (train.py:430-437),
//                         total loss, and d(total)/d(predictions) for the backward pass
// The split lets a data-parallel caller all-reduce the 8 partial sums between the two
// halves so every rank differentiates the same global-batch loss.
#include "common.h"

enum { L_EDM_COORD = 0, L_EDM_PRECOND, L_PILOT_PROP, L_DIT_ATT, L_DIT_NORM, L_E3_COORD, L_E3_ROT, L_E3_TRANS, L_TERMS };

// train.py:66-90 weights of the tensor-valued terms
__constant__ float kLossWeight[L_TERMS] = {1.0f, 0.1f, 0.2f, 0.1f, 0.05f, 1.0f, 0.1f, 0.05f};

__global__ void k_edm_prepare(const float* __restrict__ u, const float* __restrict__ pos,
                              const float* __restrict__ noise, const int32_t* __restrict__ atom_mol, int64_t N,
                              int32_t B, float root_max, float root_delta, float sigma_data,
                              float* __restrict__ coef, float* __restrict__ c_noise, float* __restrict__ noisy,
                              float* __restrict__ pre_pos) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) {  // model1.py:36-45, :27-34
    const float sg = powf(root_max + u[i] * root_delta, 7.0f);
    const float sd2 = sigma_data * sigma_data, s2 = sg * sg;
    coef[i * 4 + 0] = sd2 / (s2 + sd2);                       // c_skip
    coef[i * 4 + 1] = sg * sigma_data / sqrtf(s2 + sd2);      // c_out
    coef[i * 4 + 2] = 1.0f / sqrtf(sd2 + s2);                 // c_in
    coef[i * 4 + 3] = (s2 + sd2) / ((sg * sigma_data) * (sg * sigma_data));  // loss weight
    c_noise[i] = logf(sg) * 0.25f;
  }
  if (i < N * 3) {  // every thread recomputes its molecule's sigma: no cross-block ordering needed
    const int m = atom_mol[i / 3];
    const float sg = powf(root_max + u[m] * root_delta, 7.0f);
    const float nz = pos[i] + sg * noise[i];
    noisy[i] = nz;
    pre_pos[i] = nz * (1.0f / sqrtf(sigma_data * sigma_data + sg * sg));
  }
}

extern "C" int ligdiff_edm_prepare(const float* u, const float* pos, const float* noise, const int32_t* atom_mol,
                                   int64_t n_atoms, int32_t n_mols, float sigma_min, float sigma_max, float sigma_data,
                                   float* coef, float* c_noise, float* noisy, float* pre_pos, void* stream) {
  if (!u || !pos || !noise || !atom_mol || !coef || !c_noise || !noisy || !pre_pos || n_atoms <= 0 || n_mols <= 0)
    return LIGDIFF_ERR_ARG;
  const double rmax = pow((double)sigma_max, 1.0 / 7.0), rmin = pow((double)sigma_min, 1.0 / 7.0);
  const int64_t n = n_atoms * 3 > n_mols ? n_atoms * 3 : n_mols;
  hipLaunchKernelGGL(k_edm_prepare, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, u, pos, noise, atom_mol,
                     n_atoms, n_mols, (float)rmax, (float)(rmin - rmax), sigma_data, coef, c_noise, noisy, pre_pos);
  LD_CHECK_LAUNCH();
  return LIGDIFF_OK;
}

struct LossIn {
  const float* pos; const float* noise; const float* noisy; const float* coef;
  const float* atom_pred; const float* pos_pred; const float* prop_pred; const float* props;
  const int32_t* mol_ptr;
  int32_t B; int32_t rot_mols;
};

// One wave per molecule: sums of the per-atom terms of its atoms, its centre-of-mass term,
// its property term and (first rot_mols molecules only) the pairwise-distance term.
__global__ __launch_bounds__(256) void k_loss_partial(LossIn a, float* __restrict__ part /*[B, L_TERMS]*/) {
  const int lane = threadIdx.x & 63;
  const int mol = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (mol >= a.B) return;
  const int a0 = a.mol_ptr[mol], n = a.mol_ptr[mol + 1] - a0;
  const float c_skip = a.coef[mol * 4 + 0], c_out = a.coef[mol * 4 + 1], w = a.coef[mol * 4 + 3];
  float s0 = 0.f, s1 = 0.f, s4 = 0.f, s5 = 0.f, cx = 0.f, cy = 0.f, cz = 0.f;
  for (int i = lane; i < n; i += 64) {
    const size_t o = (size_t)(a0 + i) * 3;
    float pn2 = 0.f, dd[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float pp = a.pos_pred[o + c];
      const float den = c_skip * a.noisy[o + c] + c_out * pp;   // train.py:360-362
      const float d = den - a.pos[o + c];
      dd[c] = d;
      s5 += d * d;
      s0 += (d * d) * w;
      const float e = d - a.noise[o + c];
      s1 += e * e;
      pn2 += pp * pp;
    }
    cx += dd[0]; cy += dd[1]; cz += dd[2];
    const float nr = sqrtf(pn2) - 1.0f;
    s4 += nr * nr;
  }
  // unbiased variance of each atom's 119 logits (train.py:614)
  float s3 = 0.f;
  for (int i = 0; i < n; ++i) {
    const float* row = a.atom_pred + (size_t)(a0 + i) * LD_A;
    const float v0 = lane < LD_A ? row[lane] : 0.f;
    const float v1 = lane + 64 < LD_A ? row[lane + 64] : 0.f;
    const float mean = wave_sum(v0 + v1) * (1.0f / LD_A);
    const float d0 = lane < LD_A ? v0 - mean : 0.f, d1 = lane + 64 < LD_A ? v1 - mean : 0.f;
    s3 += wave_sum(d0 * d0 + d1 * d1) * (1.0f / (LD_A - 1));
  }
  s0 = wave_sum(s0); s1 = wave_sum(s1); s4 = wave_sum(s4); s5 = wave_sum(s5);
  cx = wave_sum(cx); cy = wave_sum(cy); cz = wave_sum(cz);
  // centre-of-mass difference (train.py:657-661): mean over atoms of (den - pos)
  const float inv_n = 1.0f / (float)(n > 0 ? n : 1);
  const float s7 = (cx * inv_n) * (cx * inv_n) + (cy * inv_n) * (cy * inv_n) + (cz * inv_n) * (cz * inv_n);
  float s2 = 0.f;
  if (a.props != nullptr && lane < LD_P) {
    const float d = a.prop_pred[(size_t)mol * LD_P + lane] - a.props[(size_t)mol * LD_P + lane];
    s2 = d * d;
  }
  s2 = wave_sum(s2);
  float s6 = 0.f;
  if (mol < a.rot_mols && n > 1) {  // train.py:642-654
    for (int pr = lane; pr < n * n; pr += 64) {
      const int i = pr / n, j = pr % n;
      if (i == j) continue;
      float dp = 0.f, dt = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const size_t oi = (size_t)(a0 + i) * 3 + c, oj = (size_t)(a0 + j) * 3 + c;
        const float di = c_skip * a.noisy[oi] + c_out * a.pos_pred[oi];
        const float dj = c_skip * a.noisy[oj] + c_out * a.pos_pred[oj];
        dp += (di - dj) * (di - dj);
        dt += (a.pos[oi] - a.pos[oj]) * (a.pos[oi] - a.pos[oj]);
      }
      const float e = sqrtf(dp) - sqrtf(dt);
      s6 += e * e;
    }
    s6 = wave_sum(s6) * (0.1f / ((float)n * (float)n));
  }
  if (lane == 0) {
    float* p = part + (size_t)mol * L_TERMS;
    p[L_EDM_COORD] = s0; p[L_EDM_PRECOND] = s1; p[L_PILOT_PROP] = s2; p[L_DIT_ATT] = s3;
    p[L_DIT_NORM] = s4; p[L_E3_COORD] = s5; p[L_E3_ROT] = s6; p[L_E3_TRANS] = s7;
  }
}

// Fixed-order reduction of the per-molecule partials (one block).
__global__ __launch_bounds__(256) void k_loss_reduce(const float* __restrict__ part, int32_t B, float* __restrict__ sums) {
  __shared__ float sh[4][L_TERMS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc[L_TERMS];
#pragma unroll
  for (int k = 0; k < L_TERMS; ++k) acc[k] = 0.f;
  for (int m = threadIdx.x; m < B; m += 256)
#pragma unroll
    for (int k = 0; k < L_TERMS; ++k) acc[k] += part[(size_t)m * L_TERMS + k];
#pragma unroll
  for (int k = 0; k < L_TERMS; ++k) {
    const float v = wave_sum(acc[k]);
    if (lane == 0) sh[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < L_TERMS) sums[threadIdx.x] = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}

// Global means, skip rule, total.  out: [0..7] terms, [8] total, [16..23] gradient
// coefficients (loss weight, or 0 for a skipped term).
__global__ void k_loss_finish(const float* __restrict__ sums, float n_atoms_g, float n_mols_g, int has_props,
                              int has_rot, float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float t[L_TERMS];
  int present[L_TERMS];
  for (int k = 0; k < L_TERMS; ++k) { t[k] = 0.f; present[k] = 1; }
  t[L_EDM_COORD] = sums[L_EDM_COORD] / (3.0f * n_atoms_g);
  t[L_EDM_PRECOND] = sums[L_EDM_PRECOND] / (3.0f * n_atoms_g) * 0.1f;
  t[L_PILOT_PROP] = sums[L_PILOT_PROP] / ((float)LD_P * n_mols_g) * 0.2f;
  t[L_DIT_ATT] = sums[L_DIT_ATT] / n_atoms_g * 0.1f;
  t[L_DIT_NORM] = sums[L_DIT_NORM] / n_atoms_g * 0.01f;
  t[L_E3_COORD] = sums[L_E3_COORD] / (3.0f * n_atoms_g);
  t[L_E3_ROT] = sums[L_E3_ROT];
  t[L_E3_TRANS] = sums[L_E3_TRANS] / (3.0f * n_mols_g) * 0.01f;
  present[L_PILOT_PROP] = has_props;
  present[L_E3_ROT] = has_rot;                 // train.py:640: only when the batch has > 1 molecule
  present[L_E3_TRANS] = n_mols_g > 1.5f;       // train.py:657
  float total = 0.f;
  for (int k = 0; k < L_TERMS; ++k) {
    const bool use = present[k] && !isnan(t[k]) && !isinf(t[k]);
    out[k] = present[k] ? t[k] : 0.f;
    out[16 + k] = use ? kLossWeight[k] : 0.f;
    if (use) total += kLossWeight[k] * t[k];
  }
  out[8] = total;
}

struct LossGradOut { float* d_atom; float* d_pos; float* d_prop; };

__global__ __launch_bounds__(256) void k_loss_grad(LossIn a, const float* __restrict__ fin, float n_atoms_g,
                                                   float n_mols_g, float upstream, LossGradOut g) {
  const int lane = threadIdx.x & 63;
  const int mol = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (mol >= a.B) return;
  const int a0 = a.mol_ptr[mol], n = a.mol_ptr[mol + 1] - a0;
  const float c_skip = a.coef[mol * 4 + 0], c_out = a.coef[mol * 4 + 1], w = a.coef[mol * 4 + 3];
  const float W0 = fin[16 + L_EDM_COORD] * upstream, W1 = fin[16 + L_EDM_PRECOND] * upstream;
  const float W2 = fin[16 + L_PILOT_PROP] * upstream, W3 = fin[16 + L_DIT_ATT] * upstream;
  const float W4 = fin[16 + L_DIT_NORM] * upstream, W5 = fin[16 + L_E3_COORD] * upstream;
  const float W6 = fin[16 + L_E3_ROT] * upstream, W7 = fin[16 + L_E3_TRANS] * upstream;
  const float inv3n = 1.0f / (3.0f * n_atoms_g);
  // centre-of-mass difference of this molecule
  float cx = 0.f, cy = 0.f, cz = 0.f;
  for (int i = lane; i < n; i += 64) {
    const size_t o = (size_t)(a0 + i) * 3;
    cx += c_skip * a.noisy[o + 0] + c_out * a.pos_pred[o + 0] - a.pos[o + 0];
    cy += c_skip * a.noisy[o + 1] + c_out * a.pos_pred[o + 1] - a.pos[o + 1];
    cz += c_skip * a.noisy[o + 2] + c_out * a.pos_pred[o + 2] - a.pos[o + 2];
  }
  const float inv_n = 1.0f / (float)(n > 0 ? n : 1);
  const float com[3] = {wave_sum(cx) * inv_n, wave_sum(cy) * inv_n, wave_sum(cz) * inv_n};
  const float ktr = W7 * 0.01f * 2.0f / (3.0f * n_mols_g) * inv_n;
  const bool rot = (mol < a.rot_mols && n > 1 && W6 != 0.f);
  for (int i = lane; i < n; i += 64) {
    const size_t o = (size_t)(a0 + i) * 3;
    float pp[3], den[3], gd[3], pn2 = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      pp[c] = a.pos_pred[o + c];
      den[c] = c_skip * a.noisy[o + c] + c_out * pp[c];
      const float d = den[c] - a.pos[o + c];
      // a term left out of the total (weight 0: absent, NaN or Inf) sends NO gradient - selected, not
      // multiplied, because its own derivative may be NaN / Inf (train.py:430-437)
      gd[c] = (W0 != 0.f ? W0 * w * 2.0f * d * inv3n : 0.f) + (W5 != 0.f ? W5 * 2.0f * d * inv3n : 0.f) +
              (W1 != 0.f ? W1 * 0.1f * 2.0f * (d - a.noise[o + c]) * inv3n : 0.f) + (W7 != 0.f ? ktr * com[c] : 0.f);
      pn2 += pp[c] * pp[c];
    }
    if (rot) {
      const float kr = W6 * 0.1f * 4.0f / ((float)n * (float)n);
      for (int j = 0; j < n; ++j) {
        if (j == i) continue;
        const size_t oj = (size_t)(a0 + j) * 3;
        float df[3], dp = 0.f, dt = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          df[c] = den[c] - (c_skip * a.noisy[oj + c] + c_out * a.pos_pred[oj + c]);
          dp += df[c] * df[c];
          dt += (a.pos[o + c] - a.pos[oj + c]) * (a.pos[o + c] - a.pos[oj + c]);
        }
        const float Dp = sqrtf(dp);
        if (Dp > 0.f) {
          const float f = kr * (Dp - sqrtf(dt)) / Dp;
#pragma unroll
          for (int c = 0; c < 3; ++c) gd[c] += f * df[c];
        }
      }
    }
    const float nr = sqrtf(pn2);
    const float kn = (nr > 0.f && W4 != 0.f) ? W4 * 0.01f * 2.0f * (nr - 1.0f) / n_atoms_g / nr : 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) g.d_pos[o + c] = gd[c] * c_out + (W4 != 0.f ? kn * pp[c] : 0.f);
  }
  const float ka = W3 * 0.1f * 2.0f / ((float)(LD_A - 1) * n_atoms_g);
  for (int i = 0; i < n; ++i) {
    const float* row = a.atom_pred + (size_t)(a0 + i) * LD_A;
    const float v0 = lane < LD_A ? row[lane] : 0.f;
    const float v1 = lane + 64 < LD_A ? row[lane + 64] : 0.f;
    const float mean = wave_sum(v0 + v1) * (1.0f / LD_A);
    float* orow = g.d_atom + (size_t)(a0 + i) * LD_A;
    if (lane < LD_A) orow[lane] = W3 != 0.f ? ka * (v0 - mean) : 0.f;
    if (lane + 64 < LD_A) orow[lane + 64] = W3 != 0.f ? ka * (v1 - mean) : 0.f;
  }
  if (g.d_prop != nullptr && lane < LD_P) {
    float v = 0.f;
    if (a.props != nullptr && W2 != 0.f)
      v = W2 * 0.2f * 2.0f * (a.prop_pred[(size_t)mol * LD_P + lane] - a.props[(size_t)mol * LD_P + lane]) /
          ((float)LD_P * n_mols_g);
    g.d_prop[(size_t)mol * LD_P + lane] = v;
  }
}

static int fill_in(LossIn* in, const ligdiff_loss_args* a) {
  if (!a || !a->pos || !a->noise || !a->noisy || !a->coef || !a->atom_pred || !a->pos_pred || !a->mol_ptr ||
      a->n_mols <= 0 || !a->workspace)
    return LIGDIFF_ERR_ARG;
  if (a->properties && !a->prop_pred) return LIGDIFF_ERR_ARG;
  in->pos = a->pos; in->noise = a->noise; in->noisy = a->noisy; in->coef = a->coef;
  in->atom_pred = a->atom_pred; in->pos_pred = a->pos_pred; in->prop_pred = a->prop_pred; in->props = a->properties;
  in->mol_ptr = a->mol_ptr; in->B = a->n_mols; in->rot_mols = a->rot_mols;
  return LIGDIFF_OK;
}

extern "C" size_t ligdiff_loss_workspace_bytes(int32_t n_mols) { return (size_t)n_mols * L_TERMS * 4 + 256; }

extern "C" int ligdiff_loss_partial(const ligdiff_loss_args* a, float* sums8, void* stream) {
  LossIn in;
  int rc = fill_in(&in, a);
  if (rc) return rc;
  if (!sums8) return LIGDIFF_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  float* part = (float*)a->workspace;
  hipLaunchKernelGGL(k_loss_partial, dim3(ceil_div(a->n_mols, 4)), dim3(256), 0, s, in, part);
  hipLaunchKernelGGL(k_loss_reduce, dim3(1), dim3(256), 0, s, part, a->n_mols, sums8);
  LD_CHECK_LAUNCH();
  return LIGDIFF_OK;
}

extern "C" int ligdiff_loss_finish(const ligdiff_loss_args* a, const float* sums8, float upstream, float* losses32,
                                   float* d_atom, float* d_pos, float* d_prop, void* stream) {
  LossIn in;
  int rc = fill_in(&in, a);
  if (rc) return rc;
  if (!sums8 || !losses32 || !d_atom || !d_pos) return LIGDIFF_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const float ng = (float)a->n_atoms_global, bg = (float)a->n_mols_global;
  hipLaunchKernelGGL(k_loss_finish, dim3(1), dim3(64), 0, s, sums8, ng, bg, a->properties != nullptr ? 1 : 0,
                     a->n_mols_global > 1 ? 1 : 0, losses32);
  LossGradOut g{d_atom, d_pos, d_prop};
  hipLaunchKernelGGL(k_loss_grad, dim3(ceil_div(a->n_mols, 4)), dim3(256), 0, s, in, losses32, ng, bg, upstream, g);
  LD_CHECK_LAUNCH();
  return LIGDIFF_OK;
}
