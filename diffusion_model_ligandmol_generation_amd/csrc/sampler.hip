// Reverse-diffusion updates (generation.py:457-484 DDPM, :359-381 DDIM) and the
// graph-captured sampling loop (generation.py:422).
#include <stdlib.h>
#include <string.h>

#include "common.h"
#include "forward.h"

// ------------------------------------------------------------------ Gaussian draws (Philox4x32-10, common.h)
// Four N(0,1) draws for the quad of elements [4q, 4q+4) of stream `sid` at step `step`.
// The counter is the GLOBAL quad index, so the draw for an element does not depend on how
// molecules are split over GPUs or launches.
__device__ __forceinline__ f32x4 philox_normal4(uint64_t seed, uint32_t step, uint32_t sid, uint64_t quad) {
  uint32_t c[4] = {(uint32_t)quad, (uint32_t)(quad >> 32), step, sid};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  const float k = 2.3283064365386963e-10f;  // 2^-32
  const float u0 = ((float)c[0] + 0.5f) * k, u1 = ((float)c[1] + 0.5f) * k;
  const float u2 = ((float)c[2] + 0.5f) * k, u3 = ((float)c[3] + 0.5f) * k;
  const float r0 = sqrtf(-2.0f * logf(fminf(u0, 0.99999994f) + 1e-30f));
  const float r1 = sqrtf(-2.0f * logf(fminf(u2, 0.99999994f) + 1e-30f));
  float s0, c0, s1, c1;
  sincosf(6.283185307179586f * u1, &s0, &c0);
  sincosf(6.283185307179586f * u3, &s1, &c1);
  return (f32x4){r0 * c0, r0 * s0, r1 * c1, r1 * s1};
}

__global__ void k_philox_normal(float* out, int64_t n, uint64_t seed, uint32_t step, uint32_t sid, int64_t offset,
                                float scale) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t g = (uint64_t)(offset + i);
  const f32x4 z = philox_normal4(seed, step, sid, g >> 2);
  out[i] = z[g & 3] * scale;
}

extern "C" int ligdiff_philox_normal(float* out, int64_t n, uint64_t seed, uint32_t step, uint32_t stream_id,
                                     int64_t offset, float scale, void* stream) {
  if (!out || n < 0 || offset < 0) return LIGDIFF_ERR_ARG;
  if (n == 0) return LIGDIFF_OK;
  hipLaunchKernelGGL(k_philox_normal, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, out, n, seed, step,
                     stream_id, offset, scale);
  LD_CHECK_LAUNCH();
  return LIGDIFF_OK;
}

// ------------------------------------------------------------------ fused step kernels
struct StepCoef {  // scalars of one reverse update: v <- a * (v - b * eps) + c * z
  float a, b, c;
  int add_noise;
};

__device__ __forceinline__ StepCoef ddpm_coef(const float* betas, const float* alphas, const float* abar, int t,
                                              float temperature) {
  StepCoef k;
  if (t > 0) {
    const float beta = betas[t];
    k.a = 1.0f / sqrtf(alphas[t]);
    k.b = beta / sqrtf(1.0f - abar[t]);
    k.c = sqrtf(beta);
    k.add_noise = t > 1;
  } else {  // final step: (v - sqrt(1-abar_0) eps) / sqrt(abar_0)
    k.a = 0.f; k.b = sqrtf(1.0f - abar[0]); k.c = sqrtf(abar[0]);
    k.add_noise = 0;
  }
  (void)temperature;
  return k;
}

// One thread per QUAD of four consecutive global elements (one Philox call yields four normals;
// a thread per element would compute each call four times).  stream id 0 = atom features,
// 1 = coordinates; the quad index is global (atom_offset included), so the first and last quad of
// an array may be partial.  Used for the DDPM update; the DDIM update draws no noise.
__global__ __launch_bounds__(256) void k_reverse_step_ddpm(ligdiff_step_args a) {
  const int64_t nx = a.n_atoms * LD_A, np = a.n_atoms * 3;
  const int64_t gx0 = a.atom_offset * LD_A, gp0 = a.atom_offset * 3;
  // quads covering [g0, g0 + n): from g0 >> 2 to (g0 + n - 1) >> 2
  const int64_t qx = ((gx0 + nx - 1) >> 2) - (gx0 >> 2) + 1;
  const int64_t qp = ((gp0 + np - 1) >> 2) - (gp0 >> 2) + 1;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= qx + qp) return;
  const bool is_pos = i >= qx;
  const int64_t g0 = is_pos ? gp0 : gx0, n = is_pos ? np : nx;
  const uint64_t quad = (uint64_t)((g0 >> 2) + (is_pos ? i - qx : i));
  float* state = is_pos ? a.pos : a.x;
  const float* ec = is_pos ? a.eps_pos : a.eps_x;
  const float* eu = is_pos ? a.eps_pos_uncond : a.eps_x_uncond;
  const float* nz = is_pos ? a.noise_pos : a.noise_x;
  const int t = *a.t_index_dev;
  const StepCoef k = ddpm_coef(a.betas, a.alphas, a.alphas_cumprod, t, a.temperature);
  f32x4 z = {0.f, 0.f, 0.f, 0.f};
  if (t > 0 && k.add_noise && nz == nullptr) z = philox_normal4(a.seed, (uint32_t)t, is_pos ? 1u : 0u, quad);
  const int64_t j0 = (int64_t)(quad * 4) - g0;
  // whole quad inside the array and 16-B aligned (always so for a single shard at offset 0): vector form
  if (j0 >= 0 && j0 + 3 < n && (j0 & 3) == 0 &&
      ((reinterpret_cast<uintptr_t>(state) | reinterpret_cast<uintptr_t>(ec) | reinterpret_cast<uintptr_t>(eu) |
        reinterpret_cast<uintptr_t>(nz)) & 15) == 0) {
    f32x4 eps = *reinterpret_cast<const f32x4*>(ec + j0);
    if (eu != nullptr) {  // generation.py:442-443
      const f32x4 u = *reinterpret_cast<const f32x4*>(eu + j0);
#pragma unroll
      for (int e = 0; e < 4; ++e) eps[e] = u[e] + a.guidance_scale * (eps[e] - u[e]);
    }
    const f32x4 v = *reinterpret_cast<const f32x4*>(state + j0);
    if (nz != nullptr && t > 0 && k.add_noise) z = *reinterpret_cast<const f32x4*>(nz + j0);
    f32x4 out;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (t > 0) {
        out[e] = k.a * (v[e] - k.b * eps[e]);
        if (k.add_noise) out[e] += (z[e] * k.c) * a.temperature;
      } else {
        out[e] = (v[e] - k.b * eps[e]) / k.c;
      }
    }
    *reinterpret_cast<f32x4*>(state + j0) = out;
    return;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int64_t j = j0 + e;   // local element
    if (j < 0 || j >= n) continue;
    float eps = ec[j];
    if (eu != nullptr) {  // generation.py:442-443
      const float u = eu[j];
      eps = u + a.guidance_scale * (eps - u);
    }
    const float v = state[j];
    float out;
    if (t > 0) {
      out = k.a * (v - k.b * eps);
      if (k.add_noise) out += ((nz != nullptr ? nz[j] : z[e]) * k.c) * a.temperature;
    } else {
      out = (v - k.b * eps) / k.c;
    }
    state[j] = out;
  }
}

// DDIM update (generation.py:359-381), one thread per element.
__global__ __launch_bounds__(256) void k_reverse_step_ddim(ligdiff_step_args a, int t_cur_arg, int t_next_arg) {
  const int64_t nx = a.n_atoms * LD_A, np = a.n_atoms * 3;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nx + np) return;
  const bool is_pos = i >= nx;
  const int64_t j = is_pos ? i - nx : i;
  float* state = is_pos ? a.pos : a.x;
  const float* ec = is_pos ? a.eps_pos : a.eps_x;
  const float* eu = is_pos ? a.eps_pos_uncond : a.eps_x_uncond;
  float eps = ec[j];
  if (eu != nullptr) {  // generation.py:442-443
    const float u = eu[j];
    eps = u + a.guidance_scale * (eps - u);
  }
  const float v = state[j];
  const float at = a.alphas_cumprod[t_cur_arg];
  const float x0 = (v - sqrtf(1.0f - at) * eps) / sqrtf(at);
  float out = x0;
  if (t_next_arg >= 0) {
    const float an = a.alphas_cumprod[t_next_arg];
    out = sqrtf(an) * x0 + sqrtf(1.0f - an) * eps;
  }
  state[j] = out;
}

// Coordinate-only update of the alternate samplers: pos <- (pos - b eps) / d + c z.
__global__ __launch_bounds__(256) void k_pos_step(float* __restrict__ pos, const float* __restrict__ ec,
                                                  const float* __restrict__ eu, float guidance_scale, int64_t n,
                                                  float b, float d, float c, const float* __restrict__ noise,
                                                  uint64_t seed, uint32_t step, int64_t elem_offset) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float eps = ec[i];
  if (eu != nullptr) {
    const float u = eu[i];
    eps = u + guidance_scale * (eps - u);
  }
  float out = (pos[i] - eps * b) / d;
  if (c != 0.0f) {
    float z;
    if (noise != nullptr) {
      z = noise[i];
    } else {
      const uint64_t g = (uint64_t)(elem_offset + i);
      z = philox_normal4(seed, step, 1u, g >> 2)[g & 3];
    }
    out += c * z;
  }
  pos[i] = out;
}

extern "C" int ligdiff_pos_step(float* pos, const float* eps_pos, const float* eps_pos_uncond, float guidance_scale,
                                int64_t n_atoms, float b, float d, float c, const float* noise_pos, uint64_t seed,
                                uint32_t step, int64_t atom_offset, void* stream) {
  if (!pos || !eps_pos || n_atoms <= 0 || atom_offset < 0 || d == 0.0f) return LIGDIFF_ERR_ARG;
  const int64_t n = n_atoms * 3;
  hipLaunchKernelGGL(k_pos_step, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, pos, eps_pos,
                     eps_pos_uncond, guidance_scale, n, b, d, c, noise_pos, seed, step, atom_offset * 3);
  LD_CHECK_LAUNCH();
  return LIGDIFF_OK;
}

__global__ void k_add_int(int32_t* p, int32_t d) { *p += d; }
__global__ void k_set_int(int32_t* p, int32_t v) { *p = v; }

static void launch_reverse_step_ddpm(const ligdiff_step_args* a, hipStream_t s) {
  const int64_t quads = (a->n_atoms * LD_A + 3) / 4 + (a->n_atoms * 3 + 3) / 4 + 2;   // upper bound incl. partial ends
  hipLaunchKernelGGL(k_reverse_step_ddpm, dim3(ceil_div(quads, 256)), dim3(256), 0, s, *a);
}

static int check_step_args(const ligdiff_step_args* a) {
  if (!a || !a->x || !a->pos || !a->eps_x || !a->eps_pos || a->n_atoms <= 0) return LIGDIFF_ERR_ARG;
  if ((a->eps_x_uncond == nullptr) != (a->eps_pos_uncond == nullptr)) return LIGDIFF_ERR_ARG;
  if ((a->noise_x == nullptr) != (a->noise_pos == nullptr)) return LIGDIFF_ERR_ARG;
  return LIGDIFF_OK;
}

extern "C" int ligdiff_ddpm_step(const ligdiff_step_args* a, void* stream) {
  int rc = check_step_args(a);
  if (rc) return rc;
  if (!a->betas || !a->alphas || !a->alphas_cumprod || !a->t_index_dev) return LIGDIFF_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  launch_reverse_step_ddpm(a, s);
  hipLaunchKernelGGL(k_add_int, dim3(1), dim3(1), 0, s, a->t_index_dev, -1);
  LD_CHECK_LAUNCH();
  return LIGDIFF_OK;
}

extern "C" int ligdiff_ddim_step(const ligdiff_step_args* a, int32_t t_cur, int32_t t_next, void* stream) {
  int rc = check_step_args(a);
  if (rc) return rc;
  if (!a->alphas_cumprod || t_cur < 0) return LIGDIFF_ERR_ARG;
  const int64_t n = a->n_atoms * (LD_A + 3);
  hipLaunchKernelGGL(k_reverse_step_ddim, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, *a, t_cur, t_next);
  LD_CHECK_LAUNCH();
  return LIGDIFF_OK;
}

// ------------------------------------------------------------------ whole DDPM loop
// The time embedding and the adaLN parameters depend on the step index only, so they are computed
// for ALL steps before the loop (n_steps rows, same kernels and per-row arithmetic as the one-row
// path); each step then copies its row into the forward's shared-timestep buffers with one small
// launch that also advances the step counter.
#define TIME_ROW_FLOATS (256 + 3 * 1536)
// Guided sampling (classifier-free guidance) evaluates the network on the conditional batch and on an
// unconditional copy of it.  Both run as ONE forward over a doubled batch (molecules [0, B) with the
// property embedding, [B, 2B) without): twice the rows per launch fills the GPU where B molecules
// alone leave most GEMMs at one workgroup per CU.  Molecules never interact, so the result equals two
// separate passes bit for bit.  LIGDIFF_CFG_TWO_PASS=1 keeps the two-pass form (A/B switch used by the tests).
struct SampleLayout {
  size_t fwd, eps_x, eps_pos, eps_xu, eps_pu, t_index, tvals, tf, t1, stemb, tab_temb, tab_mod[LD_MAXL];
  // doubled batch of the one-pass guided form: state copies and the int32 arrays of its ligdiff_graph
  size_t x2, pos2, g_mol_ptr, g_atom_mol, g_dst_ptr, g_src_ptr, g_src, g_dst, g_src_slot;
  size_t total;
};
static void sample_layout(SampleLayout* L, int64_t N, int64_t E, int32_t B, int guided, int32_t n_steps) {
  memset(L, 0, sizeof *L);
  const bool one_pass = guided && cfg_one_pass();
  const int64_t m = one_pass ? 2 : 1;   // batch multiplier of the forward
  FwdLayout F;
  fwd_layout(&F, m * N, m * E, (int32_t)(m * B), LIGDIFF_FWD_T_BROADCAST);
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes, 256); return o; };
  L->fwd = take(F.total);
  if (one_pass) {   // conditional and unconditional predictions are the two halves of one output
    L->eps_x = take((size_t)2 * N * LD_A * 4); L->eps_pos = take((size_t)2 * N * 3 * 4);
    L->eps_xu = L->eps_x + (size_t)N * LD_A * 4;
    L->eps_pu = L->eps_pos + (size_t)N * 3 * 4;
    L->x2 = take((size_t)2 * N * LD_A * 4); L->pos2 = take((size_t)2 * N * 3 * 4);
    L->g_mol_ptr = take((size_t)(2 * B + 1) * 4); L->g_atom_mol = take((size_t)2 * N * 4);
    L->g_dst_ptr = take((size_t)(2 * N + 1) * 4); L->g_src_ptr = take((size_t)(2 * N + 1) * 4);
    L->g_src = take((size_t)2 * E * 4 + 4); L->g_dst = take((size_t)2 * E * 4 + 4); L->g_src_slot = take((size_t)2 * E * 4 + 4);
  } else {
    L->eps_x = take((size_t)N * LD_A * 4); L->eps_pos = take((size_t)N * 3 * 4);
    L->eps_xu = guided ? take((size_t)N * LD_A * 4) : 0;
    L->eps_pu = guided ? take((size_t)N * 3 * 4) : 0;
  }
  L->t_index = take(256);
  const size_t T = (size_t)(n_steps > 0 ? n_steps : 1);
  L->tvals = take(T * 4); L->tf = take(T * 256 * 4); L->t1 = take(T * 512 * 4); L->stemb = take(T * 256 * 4);
  L->tab_temb = take(T * 256 * 4);
  for (int l = 0; l < F.nl; ++l) L->tab_mod[l] = take(T * 1536 * 4);
  L->total = off;
}

extern "C" size_t ligdiff_ddpm_sample_workspace_bytes(int64_t n_atoms, int64_t n_edges, int32_t n_mols, int guided,
                                                      int32_t n_steps) {
  SampleLayout L;
  sample_layout(&L, n_atoms, n_edges, n_mols, guided, n_steps);
  return L.total;
}

__global__ void k_iota_float(float* v, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = (float)i;
}

// Step prologue, ONE workgroup: row r = counter[0] of the tables -> the forward's temb / mod buffers
// (1216 float4), then counter[1] = r (index of the step now in progress, read by its update
// kernel) and counter[0] = r + delta (the next one).  A single workgroup so that the counter can be
// advanced after its barrier: every thread has read counter[0] by then.
struct TimeGatherArgs { const float* tab_temb; const float* tab_mod[LD_MAXL]; float* temb; float* mod[LD_MAXL]; int n_layers; };
__global__ __launch_bounds__(1024) void k_step_prologue(int32_t* counter, TimeGatherArgs p, int delta) {
  const int t = counter[0];
  for (int i = threadIdx.x; i < 64 + p.n_layers * 384; i += 1024) {
    if (i < 64) {
      reinterpret_cast<f32x4*>(p.temb)[i] = reinterpret_cast<const f32x4*>(p.tab_temb + (size_t)t * 256)[i];
    } else {
      const int l = (i - 64) / 384, j = (i - 64) % 384;
      reinterpret_cast<f32x4*>(p.mod[l])[j] = reinterpret_cast<const f32x4*>(p.tab_mod[l] + (size_t)t * 1536)[j];
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    counter[1] = t;
    counter[0] = t + delta;
  }
}
__global__ void k_int_to_float(const int32_t* in, float* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (float)in[i];
}

// DDIM update (generation.py:359-381) for loop position i = *pos_dev of the timestep list ts[n_steps]:
// t_cur = ts[i], t_next = ts[i+1] (none after the last).  One thread per element.
__global__ __launch_bounds__(256) void k_reverse_step_ddim_list(ligdiff_step_args a, const int32_t* __restrict__ ts,
                                                                int n_steps, const int32_t* __restrict__ pos_dev) {
  const int64_t nx = a.n_atoms * LD_A, np = a.n_atoms * 3;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nx + np) return;
  const int step = *pos_dev;
  const int t_cur = ts[step];
  const int t_next = step + 1 < n_steps ? ts[step + 1] : -1;
  const bool is_pos = i >= nx;
  const int64_t j = is_pos ? i - nx : i;
  float* state = is_pos ? a.pos : a.x;
  const float* ec = is_pos ? a.eps_pos : a.eps_x;
  const float* eu = is_pos ? a.eps_pos_uncond : a.eps_x_uncond;
  float eps = ec[j];
  if (eu != nullptr) {
    const float u = eu[j];
    eps = u + a.guidance_scale * (eps - u);
  }
  const float at = a.alphas_cumprod[t_cur];
  const float x0 = (state[j] - sqrtf(1.0f - at) * eps) / sqrtf(at);
  float out = x0;
  if (t_next >= 0) {
    const float an = a.alphas_cumprod[t_next];
    out = sqrtf(an) * x0 + sqrtf(1.0f - an) * eps;
  }
  state[j] = out;
}

// Per-step prediction gate of the DDIM loop on the guidance-combined prediction
// (generation.py:351, :442-443, :518-541).
__global__ void k_prediction_gate_guided(ligdiff_step_args a, const int32_t* __restrict__ atom_mol,
                                         int32_t* __restrict__ reject) {
  const int64_t nx = a.n_atoms * LD_A, np = a.n_atoms * 3;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nx + np) return;
  const bool is_pos = i >= nx;
  const int64_t j = is_pos ? i - nx : i;
  float v = (is_pos ? a.eps_pos : a.eps_x)[j];
  const float* eu = is_pos ? a.eps_pos_uncond : a.eps_x_uncond;
  if (eu != nullptr) {
    const float u = eu[j];
    v = u + a.guidance_scale * (v - u);
  }
  const int64_t atom = is_pos ? j / 3 : j / LD_A;
  if (!(fabsf(v) <= 100.0f)) reject[atom_mol[atom]] = 1;  // also true for NaN
}

// Arrays of the doubled batch: the second copy of every molecule, atom and edge carries index + B / N / E.
struct DoubleGraphArgs {
  const int32_t *mol_ptr, *atom_mol, *dst_ptr, *src_ptr, *src, *dst, *src_slot;
  int32_t *mol_ptr2, *atom_mol2, *dst_ptr2, *src_ptr2, *src2, *dst2, *src_slot2;
  int N, E, B;
};
__global__ void k_double_graph(DoubleGraphArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i <= a.B) { a.mol_ptr2[i] = a.mol_ptr[i]; a.mol_ptr2[a.B + i] = a.mol_ptr[i] + a.N; }
  if (i <= a.N) {
    a.dst_ptr2[i] = a.dst_ptr[i]; a.dst_ptr2[a.N + i] = a.dst_ptr[i] + a.E;
    a.src_ptr2[i] = a.src_ptr[i]; a.src_ptr2[a.N + i] = a.src_ptr[i] + a.E;
  }
  if (i < a.N) { a.atom_mol2[i] = a.atom_mol[i]; a.atom_mol2[a.N + i] = a.atom_mol[i] + a.B; }
  if (i < a.E) {
    a.src2[i] = a.src[i]; a.src2[a.E + i] = a.src[i] + a.N;
    a.dst2[i] = a.dst[i]; a.dst2[a.E + i] = a.dst[i] + a.N;
    a.src_slot2[i] = a.src_slot[i]; a.src_slot2[a.E + i] = a.src_slot[i] + a.E;
  }
}
// x2 = [x; x], pos2 = [pos; pos]  (nx, np: element counts of x and pos)
__global__ void k_dup_state(const float* __restrict__ x, const float* __restrict__ pos, float* __restrict__ x2,
                            float* __restrict__ pos2, int64_t nx, int64_t np) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nx) { const float v = x[i]; x2[i] = v; x2[nx + i] = v; }
  else if (i < nx + np) { const float v = pos[i - nx]; pos2[i - nx] = v; pos2[np + i - nx] = v; }
}

// Shared loop of the DDPM sampler (ddim_ts == NULL: step index counts down from n_steps - 1 and is the
// timestep) and of the DDIM sampler (ddim_ts [n_steps] device int32: loop position counts up, the
// timestep is ddim_ts[position]; `reject` [n_mols] collects the per-step prediction gate).
static int sample_loop(const float* const* weights, const ligdiff_graph* graph, float* x, float* pos,
                       const float* properties, float guidance_scale, float temperature, const float* betas,
                       const float* alphas, const float* alphas_cumprod, int32_t n_steps, uint64_t seed,
                       int64_t atom_offset, int use_graph, void* workspace, size_t workspace_bytes, void* stream,
                       const int32_t* ddim_ts, int32_t* reject) {
  const bool ddim = ddim_ts != nullptr;
  if (!weights || !graph || !x || !pos || !alphas_cumprod || !workspace || n_steps <= 0) return LIGDIFF_ERR_ARG;
  {
    const int grc = check_graph(graph);   // also: no molecule beyond the attention kernel's 128 atoms
    if (grc) return grc;
  }
  if (!ddim && (!betas || !alphas)) return LIGDIFF_ERR_ARG;
  if (ddim && !reject) return LIGDIFF_ERR_ARG;
  if (use_graph && stream == nullptr) return LIGDIFF_ERR_ARG;  // the default stream cannot be captured
  // generation.py:428: guidance only when properties are given AND scale > 1
  const int guided = (properties != nullptr && guidance_scale > 1.0f) ? 1 : 0;
  SampleLayout L;
  sample_layout(&L, graph->n_atoms, graph->n_edges, graph->n_mols, guided, n_steps);
  if (workspace_bytes < L.total) return LIGDIFF_ERR_WORKSPACE;
  char* ws = (char*)workspace;
  hipStream_t s = (hipStream_t)stream;
  int32_t* counter = (int32_t*)(ws + L.t_index);   // [0] next step index, [1] index of the step in progress
  auto WF = [&](size_t off) { return (float*)(ws + off); };

  // table rows: DDPM row t <-> timestep t; DDIM row i <-> timestep ddim_ts[i]
  if (ddim)
    hipLaunchKernelGGL(k_int_to_float, dim3(ceil_div(n_steps, 256)), dim3(256), 0, s, ddim_ts, WF(L.tvals), n_steps);
  else
    hipLaunchKernelGGL(k_iota_float, dim3(ceil_div(n_steps, 256)), dim3(256), 0, s, WF(L.tvals), n_steps);
  const int NL = call_layer_pairs();
  float* tab_mods[LD_MAXL];
  for (int l = 0; l < NL; ++l) tab_mods[l] = WF(L.tab_mod[l]);
  int rc = run_time_rows(weights, WF(L.tvals), n_steps, WF(L.tf), WF(L.t1), WF(L.tab_temb), WF(L.stemb), tab_mods, NL, s);
  if (rc) return rc;
  const bool one_pass = guided && cfg_one_pass();
  const int64_t N1 = graph->n_atoms, E1 = graph->n_edges;
  const int32_t B1 = graph->n_mols;
  if (one_pass && (2 * N1 >= (1LL << 30) || 2 * E1 >= (1LL << 30))) return LIGDIFF_ERR_ARG;
  ligdiff_graph graph2 = *graph;
  if (one_pass) {
    auto WI = [&](size_t off) { return (int32_t*)(ws + off); };
    DoubleGraphArgs d{graph->mol_ptr, graph->atom_mol, graph->dst_ptr, graph->src_ptr, graph->src, graph->dst,
                      graph->src_slot, WI(L.g_mol_ptr), WI(L.g_atom_mol), WI(L.g_dst_ptr), WI(L.g_src_ptr), WI(L.g_src),
                      WI(L.g_dst), WI(L.g_src_slot), (int)N1, (int)E1, (int)B1};
    const int64_t span = (N1 > E1 ? N1 : E1) + 1;
    hipLaunchKernelGGL(k_double_graph, dim3(ceil_div(span > B1 + 1 ? span : B1 + 1, 256)), dim3(256), 0, s, d);
    graph2.n_atoms = 2 * N1; graph2.n_edges = 2 * E1; graph2.n_mols = 2 * B1;
    graph2.mol_ptr = d.mol_ptr2; graph2.atom_mol = d.atom_mol2; graph2.dst_ptr = d.dst_ptr2; graph2.src_ptr = d.src_ptr2;
    graph2.src = d.src2; graph2.dst = d.dst2; graph2.src_slot = d.src_slot2; graph2.perm = nullptr;
  }
  FwdLayout FL;
  fwd_layout(&FL, graph2.n_atoms, graph2.n_edges, graph2.n_mols, LIGDIFF_FWD_T_BROADCAST);
  TimeGatherArgs tg;
  memset(&tg, 0, sizeof tg);
  tg.tab_temb = WF(L.tab_temb); tg.temb = WF(L.fwd + FL.temb); tg.n_layers = NL;
  for (int l = 0; l < NL; ++l) { tg.tab_mod[l] = WF(L.tab_mod[l]); tg.mod[l] = WF(L.fwd + FL.mod[l]); }

  ligdiff_step_args a;
  memset(&a, 0, sizeof a);
  a.x = x; a.pos = pos;
  a.eps_x = WF(L.eps_x); a.eps_pos = WF(L.eps_pos);
  if (guided) { a.eps_x_uncond = WF(L.eps_xu); a.eps_pos_uncond = WF(L.eps_pu); }
  a.guidance_scale = guidance_scale; a.temperature = temperature;
  a.betas = betas; a.alphas = alphas; a.alphas_cumprod = alphas_cumprod;
  a.t_index_dev = counter + 1; a.seed = seed; a.atom_offset = atom_offset; a.n_atoms = graph->n_atoms;

  const int base = LIGDIFF_FWD_T_BROADCAST | LIGDIFF_FWD_SKIP_PROP_HEAD | FWD_INTERNAL_TIME_GIVEN;
  SplitCache split;   // split-operand mode: weight blocks converted once for the whole loop
  auto forwards = [&]() -> int {
    if (one_pass) {
      const int64_t nx = N1 * LD_A, np = N1 * 3;
      hipLaunchKernelGGL(k_dup_state, dim3(ceil_div(nx + np, 256)), dim3(256), 0, s, x, pos, WF(L.x2), WF(L.pos2), nx, np);
      return run_forward(weights, &graph2, WF(L.x2), WF(L.pos2), nullptr, counter + 1, properties,
                         base | LIGDIFF_FWD_PROPERTIES | FWD_INTERNAL_HALF_COND, WF(L.eps_x), WF(L.eps_pos), nullptr,
                         ws + L.fwd, s, nullptr, &split);
    }
    int r = run_forward(weights, graph, x, pos, nullptr, counter + 1, properties,
                        base | (properties ? LIGDIFF_FWD_PROPERTIES : 0), WF(L.eps_x), WF(L.eps_pos), nullptr,
                        ws + L.fwd, s, nullptr, &split);
    if (r) return r;
    if (guided)
      r = run_forward(weights, graph, x, pos, nullptr, counter + 1, nullptr, base, WF(L.eps_xu), WF(L.eps_pu), nullptr,
                      ws + L.fwd, s, nullptr, &split);
    return r;
  };
  auto one_step = [&]() -> int {
    hipLaunchKernelGGL(k_step_prologue, dim3(1), dim3(1024), 0, s, counter, tg, ddim ? 1 : -1);
    int r = forwards();
    if (r) return r;
    if (ddim) {
      const int64_t n = a.n_atoms * (LD_A + 3);
      hipLaunchKernelGGL(k_prediction_gate_guided, dim3(ceil_div(n, 256)), dim3(256), 0, s, a, graph->atom_mol, reject);
      hipLaunchKernelGGL(k_reverse_step_ddim_list, dim3(ceil_div(n, 256)), dim3(256), 0, s, a, ddim_ts, n_steps,
                         (const int32_t*)(counter + 1));
    } else {
      launch_reverse_step_ddpm(&a, s);
    }
    return hipGetLastError() == hipSuccess ? LIGDIFF_OK : LIGDIFF_ERR_LAUNCH;
  };

  hipLaunchKernelGGL(k_set_int, dim3(1), dim3(1), 0, s, counter, ddim ? 0 : n_steps - 1);
  if (use_graph && split_gemm_enabled()) {
    // one forward outside the capture (its outputs are overwritten by the first step) so that the
    // weight conversions are not part of the replayed graphs; delta 0 leaves the counter where it is
    hipLaunchKernelGGL(k_step_prologue, dim3(1), dim3(1024), 0, s, counter, tg, 0);
    rc = forwards();
    if (rc) return rc;
  }
  if (!use_graph) {
    for (int i = 0; i < n_steps; ++i) {
      rc = one_step();
      if (rc) return rc;
    }
    return LIGDIFF_OK;
  }
  // The step counter lives in device memory and is advanced by the captured kernels themselves, so
  // one executable graph serves every step.  Four steps are captured per graph (fewer graph launch
  // boundaries, each worth ~8 us of idle GPU); a one-step graph runs the remainder.
  const int kGroup = 4;
  auto capture = [&](int steps, hipGraphExec_t* exec_out) -> int {
    hipGraph_t gh = nullptr;
    if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) return LIGDIFF_ERR_LAUNCH;
    int r = LIGDIFF_OK;
    for (int i = 0; i < steps && r == LIGDIFF_OK; ++i) r = one_step();
    const hipError_t he = hipStreamEndCapture(s, &gh);
    if (r || he != hipSuccess || gh == nullptr) {
      if (gh) (void)hipGraphDestroy(gh);
      return r ? r : LIGDIFF_ERR_LAUNCH;
    }
    const hipError_t hi = hipGraphInstantiate(exec_out, gh, nullptr, nullptr, 0);
    (void)hipGraphDestroy(gh);
    return hi == hipSuccess ? LIGDIFF_OK : LIGDIFF_ERR_LAUNCH;
  };
  hipGraphExec_t exec4 = nullptr, exec1 = nullptr;
  const int groups = n_steps / kGroup, rest = n_steps % kGroup;
  rc = LIGDIFF_OK;
  if (groups > 0) rc = capture(kGroup, &exec4);
  if (rc == LIGDIFF_OK && rest > 0) rc = capture(1, &exec1);
  for (int i = 0; i < groups && rc == LIGDIFF_OK; ++i)
    if (hipGraphLaunch(exec4, s) != hipSuccess) rc = LIGDIFF_ERR_LAUNCH;
  for (int i = 0; i < rest && rc == LIGDIFF_OK; ++i)
    if (hipGraphLaunch(exec1, s) != hipSuccess) rc = LIGDIFF_ERR_LAUNCH;
  // the executable graphs must outlive their queued launches
  if (hipStreamSynchronize(s) != hipSuccess && rc == LIGDIFF_OK) rc = LIGDIFF_ERR_LAUNCH;
  if (exec4) (void)hipGraphExecDestroy(exec4);
  if (exec1) (void)hipGraphExecDestroy(exec1);
  return rc;
}

extern "C" int ligdiff_ddpm_sample_opt(const float* const* weights, const ligdiff_graph* graph, float* x, float* pos,
                                       const float* properties, float guidance_scale, float temperature,
                                       const float* betas, const float* alphas, const float* alphas_cumprod,
                                       int32_t n_steps, uint64_t seed, int64_t atom_offset, int use_graph,
                                       void* workspace, size_t workspace_bytes, void* stream, const ligdiff_options* options) {
  if (check_options(options)) return LIGDIFF_ERR_ARG;
  CallOptions scope(options);
  return ligdiff_ddpm_sample(weights, graph, x, pos, properties, guidance_scale, temperature, betas, alphas, alphas_cumprod,
                             n_steps, seed, atom_offset, use_graph, workspace, workspace_bytes, stream);
}
extern "C" size_t ligdiff_ddpm_sample_workspace_bytes_opt(int64_t n_atoms, int64_t n_edges, int32_t n_mols, int guided,
                                                          int32_t n_steps, const ligdiff_options* options) {
  if (check_options(options)) return 0;
  CallOptions scope(options);
  return ligdiff_ddpm_sample_workspace_bytes(n_atoms, n_edges, n_mols, guided, n_steps);
}
extern "C" int ligdiff_ddim_sample_opt(const float* const* weights, const ligdiff_graph* graph, float* x, float* pos,
                                       const float* properties, float guidance_scale, const float* alphas_cumprod,
                                       const int32_t* timesteps, int32_t n_steps, int32_t* reject, int use_graph,
                                       void* workspace, size_t workspace_bytes, void* stream, const ligdiff_options* options) {
  if (check_options(options)) return LIGDIFF_ERR_ARG;
  CallOptions scope(options);
  return ligdiff_ddim_sample(weights, graph, x, pos, properties, guidance_scale, alphas_cumprod, timesteps, n_steps, reject,
                             use_graph, workspace, workspace_bytes, stream);
}

extern "C" int ligdiff_ddpm_sample(const float* const* weights, const ligdiff_graph* graph, float* x, float* pos,
                                   const float* properties, float guidance_scale, float temperature,
                                   const float* betas, const float* alphas, const float* alphas_cumprod,
                                   int32_t n_steps, uint64_t seed, int64_t atom_offset, int use_graph,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  if (!betas || !alphas) return LIGDIFF_ERR_ARG;
  return sample_loop(weights, graph, x, pos, properties, guidance_scale, temperature, betas, alphas, alphas_cumprod,
                     n_steps, seed, atom_offset, use_graph, workspace, workspace_bytes, stream, nullptr, nullptr);
}

extern "C" int ligdiff_ddim_sample(const float* const* weights, const ligdiff_graph* graph, float* x, float* pos,
                                   const float* properties, float guidance_scale, const float* alphas_cumprod,
                                   const int32_t* timesteps, int32_t n_steps, int32_t* reject, int use_graph,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  if (!timesteps || !reject) return LIGDIFF_ERR_ARG;
  return sample_loop(weights, graph, x, pos, properties, guidance_scale, 1.0f, nullptr, nullptr, alphas_cumprod, n_steps,
                     0, 0, use_graph, workspace, workspace_bytes, stream, timesteps, reject);
}

// ------------------------------------------------------------------ acceptance gates
// Per-step prediction gate of the DDIM loop, generation.py:351 / :518-541: a molecule is
// rejected once any of its predictions is NaN/Inf or exceeds 100 in magnitude.
__global__ void k_prediction_gate(const float* __restrict__ eps_x, const float* __restrict__ eps_pos,
                                  const int32_t* __restrict__ atom_mol, int64_t n_atoms, int32_t* __restrict__ reject) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_atoms * (LD_A + 3)) return;
  const int64_t nx = n_atoms * LD_A;
  const float v = i < nx ? eps_x[i] : eps_pos[i - nx];
  const int64_t atom = i < nx ? i / LD_A : (i - nx) / 3;
  if (!(fabsf(v) <= 100.0f)) reject[atom_mol[atom]] = 1;  // also true for NaN
}

extern "C" int ligdiff_prediction_gate(const float* eps_x, const float* eps_pos, const int32_t* atom_mol,
                                       int64_t n_atoms, int32_t* reject, void* stream) {
  if (!eps_x || !eps_pos || !atom_mol || !reject || n_atoms <= 0) return LIGDIFF_ERR_ARG;
  hipLaunchKernelGGL(k_prediction_gate, dim3(ceil_div(n_atoms * (LD_A + 3), 256)), dim3(256), 0, (hipStream_t)stream,
                     eps_x, eps_pos, atom_mol, n_atoms, reject);
  LD_CHECK_LAUNCH();
  return LIGDIFF_OK;
}

// Final validity gates, generation.py:162-233, one wave per molecule:
//   finite values; softmax(x) max-probability >= 0.1 for every atom; unbiased per-axis
//   coordinate variance in [1e-6, 100]; every pairwise distance in [0.5, 20].
// accept[m] = 1 if the molecule passes.
__global__ __launch_bounds__(256) void k_validity_gate(const float* __restrict__ x, const float* __restrict__ pos,
                                                       const int32_t* __restrict__ mol_ptr, int32_t B,
                                                       int32_t* __restrict__ accept) {
  const int lane = threadIdx.x & 63;
  const int mol = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (mol >= B) return;
  const int a0 = mol_ptr[mol], n = mol_ptr[mol + 1] - a0;
  int bad = 0;
  // atom features: finite + dominant type
  for (int a = 0; a < n; ++a) {
    const float* row = x + (size_t)(a0 + a) * LD_A;
    const float v0 = lane < LD_A ? row[lane] : -INFINITY;
    const float v1 = lane + 64 < LD_A ? row[lane + 64] : -INFINITY;
    if ((lane < LD_A && !(fabsf(v0) <= 3.0e38f)) || (lane + 64 < LD_A && !(fabsf(v1) <= 3.0e38f))) bad = 1;
    const float mx = wave_max(fmaxf(v0, v1));
    const float s = wave_sum((lane < LD_A ? expf(v0 - mx) : 0.f) + (lane + 64 < LD_A ? expf(v1 - mx) : 0.f));
    if (!(1.0f / s >= 0.1f)) bad = 1;  // max probability = exp(0) / sum
  }
  // coordinates: finite, variance, pairwise distances
  float sx = 0.f, sy = 0.f, sz = 0.f;
  for (int a = lane; a < n; a += 64) {
    const float* p = pos + (size_t)(a0 + a) * 3;
    if (!(fabsf(p[0]) <= 3.0e38f) || !(fabsf(p[1]) <= 3.0e38f) || !(fabsf(p[2]) <= 3.0e38f)) bad = 1;
    sx += p[0]; sy += p[1]; sz += p[2];
  }
  const float inv = 1.0f / (float)n;
  const float mx = wave_sum(sx) * inv, my = wave_sum(sy) * inv, mz = wave_sum(sz) * inv;
  float qx = 0.f, qy = 0.f, qz = 0.f;
  for (int a = lane; a < n; a += 64) {
    const float* p = pos + (size_t)(a0 + a) * 3;
    qx += (p[0] - mx) * (p[0] - mx); qy += (p[1] - my) * (p[1] - my); qz += (p[2] - mz) * (p[2] - mz);
  }
  const float dn = 1.0f / (float)(n - 1);  // torch.var default: unbiased (NaN for n == 1)
  const float vx = wave_sum(qx) * dn, vy = wave_sum(qy) * dn, vz = wave_sum(qz) * dn;
  // torch.any(var < 1e-6) / torch.any(var > 100): comparisons with NaN are false
  if (vx < 1e-6f || vy < 1e-6f || vz < 1e-6f || vx > 100.f || vy > 100.f || vz > 100.f) bad = 1;
  if (n > 1) {
    for (int pr = lane; pr < n * n; pr += 64) {
      const int i = pr / n, j = pr % n;
      if (i == j) continue;
      const float* p = pos + (size_t)(a0 + i) * 3;
      const float* q = pos + (size_t)(a0 + j) * 3;
      const float dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];
      const float d = sqrtf(dx * dx + dy * dy + dz * dz);
      if (d < 0.5f || d > 20.f) bad = 1;
    }
  }
  bad = __any(bad);
  if (lane == 0) accept[mol] = bad ? 0 : 1;
}

extern "C" int ligdiff_validity_gate(const float* x, const float* pos, const int32_t* mol_ptr, int32_t n_mols,
                                     int32_t* accept, void* stream) {
  if (!x || !pos || !mol_ptr || !accept || n_mols <= 0) return LIGDIFF_ERR_ARG;
  hipLaunchKernelGGL(k_validity_gate, dim3(ceil_div(n_mols, 4)), dim3(256), 0, (hipStream_t)stream, x, pos, mol_ptr,
                     n_mols, accept);
  LD_CHECK_LAUNCH();
  return LIGDIFF_OK;
}
