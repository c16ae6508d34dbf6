// Ragged batch -> CSR packing on device (integer, deterministic, bit-exact).
//
// Edges are grouped by destination atom with their original order kept inside each group
// (the order a sequential index_add_ would visit them, model1.py:100 / :110-112), and a
// second index lists, per source atom, the CSR slots of the edges that leave it.
#include "common.h"

__global__ void k_pack_zero(int32_t* buf, int64_t n, int32_t* status) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) buf[i] = 0;
  if (i == 0) { status[0] = 0; status[1] = 0; }
}

// atom_mol = (int32) batch, sortedness / range check, mol_ptr boundaries.
__global__ void k_pack_mols(const int64_t* __restrict__ batch, int64_t N, int32_t B, int32_t* __restrict__ atom_mol,
                            int32_t* __restrict__ mol_ptr, int32_t* __restrict__ status) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int64_t b = batch[i];
  if (b < 0 || b >= B) { atomicOr(&status[0], LIGDIFF_ERR_BATCH); return; }
  atom_mol[i] = (int32_t)b;
  const int64_t prev = (i == 0) ? -1 : batch[i - 1];
  if (prev > b) { atomicOr(&status[0], LIGDIFF_ERR_BATCH); return; }
  for (int64_t m = prev + 1; m <= b; ++m) mol_ptr[m] = (int32_t)i;  // also covers empty molecules
  if (i == N - 1)
    for (int64_t m = b + 1; m <= B; ++m) mol_ptr[m] = (int32_t)N;
}

__global__ void k_pack_maxmol(const int32_t* __restrict__ mol_ptr, int32_t B, int32_t* __restrict__ status) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) atomicMax(&status[1], mol_ptr[i + 1] - mol_ptr[i]);
}

__global__ void k_pack_count(const int64_t* __restrict__ ei, int64_t E, int64_t N, int32_t* __restrict__ deg,
                             int32_t* __restrict__ sdeg, int32_t* __restrict__ status) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int64_t s = ei[e], d = ei[E + e];
  if (s < 0 || s >= N || d < 0 || d >= N) { atomicOr(&status[0], LIGDIFF_ERR_BATCH); return; }
  atomicAdd(&deg[d], 1);
  atomicAdd(&sdeg[s], 1);
}

// Exclusive scan of cnt[0..n) into ptr[0..n]; one 1024-thread block walks the array.
__global__ __launch_bounds__(1024) void k_pack_scan(const int32_t* __restrict__ cnt, int64_t n, int32_t* __restrict__ ptr) {
  __shared__ int32_t wsum[16];
  __shared__ int32_t carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int64_t base = 0; base < n; base += 1024) {
    const int64_t i = base + tid;
    const int32_t v = (i < n) ? cnt[i] : 0;
    int32_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int32_t y = __shfl_up(x, o);
      if (lane >= o) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    int32_t woff = 0;
    for (int w2 = 0; w2 < wave; ++w2) woff += wsum[w2];
    const int32_t c = carry;
    if (i < n) ptr[i] = c + woff + x - v;
    __syncthreads();
    if (tid == 1023) carry = c + woff + x;
    __syncthreads();
  }
  if (tid == 0) ptr[n] = carry;
}

// Drop item i into the bucket of key[i] (order inside a bucket fixed afterwards).
__global__ void k_pack_fill_dst(const int64_t* __restrict__ ei, int64_t E, const int32_t* __restrict__ ptr,
                                int32_t* __restrict__ cursor, int32_t* __restrict__ perm, const int32_t* status) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E || status[0] != 0) return;
  const int32_t d = (int32_t)ei[E + e];
  perm[ptr[d] + atomicAdd(&cursor[d], 1)] = (int32_t)e;
}

// Ascending sort of every bucket: restores the original (stable) order.
__global__ void k_pack_sort_buckets(const int32_t* __restrict__ ptr, int32_t* __restrict__ items, int64_t N,
                                    const int32_t* status) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N || status[0] != 0) return;
  const int32_t b = ptr[i], e = ptr[i + 1];
  for (int32_t a = b + 1; a < e; ++a) {
    const int32_t v = items[a];
    int32_t j = a - 1;
    while (j >= b && items[j] > v) { items[j + 1] = items[j]; --j; }
    items[j + 1] = v;
  }
}

__global__ void k_pack_emit(const int64_t* __restrict__ ei, int64_t E, const int32_t* __restrict__ perm,
                            int32_t* __restrict__ src, int32_t* __restrict__ dst, const int32_t* __restrict__ src_ptr,
                            int32_t* __restrict__ cursor, int32_t* __restrict__ src_slot, const int32_t* status) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= E || status[0] != 0) return;
  const int32_t e = perm[k];
  const int32_t s = (int32_t)ei[e], d = (int32_t)ei[E + e];
  src[k] = s;
  dst[k] = d;
  src_slot[src_ptr[s] + atomicAdd(&cursor[s], 1)] = (int32_t)k;
}

extern "C" size_t ligdiff_pack_workspace_bytes(int64_t n_atoms, int64_t n_edges) {
  (void)n_edges;
  return (size_t)(4 * (n_atoms + 1)) * sizeof(int32_t);
}

// Shared body of the two packing entry points; `counters` holds 4 (N + 1) int32 (degree counts and
// fill cursors of the by-destination and by-source lists), zeroed here.
static int pack_csr_run(const int64_t* edge_index, int64_t E, const int64_t* batch, int64_t N, int32_t B,
                        int32_t* mol_ptr, int32_t* atom_mol, int32_t* dst_ptr, int32_t* src, int32_t* dst,
                        int32_t* perm, int32_t* src_ptr, int32_t* src_slot, int32_t* status_dev,
                        int32_t* counters, hipStream_t s) {
  int32_t* deg = counters;
  int32_t* sdeg = deg + (N + 1);
  int32_t* cur = sdeg + (N + 1);
  int32_t* cur2 = cur + (N + 1);
  const int64_t nz = 4 * (N + 1);
  hipLaunchKernelGGL(k_pack_zero, dim3(ceil_div(nz, 256)), dim3(256), 0, s, deg, nz, status_dev);
  hipLaunchKernelGGL(k_pack_mols, dim3(ceil_div(N, 256)), dim3(256), 0, s, batch, N, B, atom_mol, mol_ptr, status_dev);
  hipLaunchKernelGGL(k_pack_maxmol, dim3(ceil_div(B, 256)), dim3(256), 0, s, mol_ptr, B, status_dev);
  if (E > 0)
    hipLaunchKernelGGL(k_pack_count, dim3(ceil_div(E, 256)), dim3(256), 0, s, edge_index, E, N, deg, sdeg, status_dev);
  hipLaunchKernelGGL(k_pack_scan, dim3(1), dim3(1024), 0, s, deg, N, dst_ptr);
  hipLaunchKernelGGL(k_pack_scan, dim3(1), dim3(1024), 0, s, sdeg, N, src_ptr);
  if (E > 0) {
    hipLaunchKernelGGL(k_pack_fill_dst, dim3(ceil_div(E, 256)), dim3(256), 0, s, edge_index, E, dst_ptr, cur, perm, status_dev);
    hipLaunchKernelGGL(k_pack_sort_buckets, dim3(ceil_div(N, 256)), dim3(256), 0, s, dst_ptr, perm, N, status_dev);
    hipLaunchKernelGGL(k_pack_emit, dim3(ceil_div(E, 256)), dim3(256), 0, s, edge_index, E, perm, src, dst, src_ptr, cur2,
                       src_slot, status_dev);
    hipLaunchKernelGGL(k_pack_sort_buckets, dim3(ceil_div(N, 256)), dim3(256), 0, s, src_ptr, src_slot, N, status_dev);
  }
  LD_CHECK_LAUNCH();
  return LIGDIFF_OK;
}

static int pack_csr_check(const int64_t* edge_index, int64_t E, const int64_t* batch, int64_t N, int32_t B,
                          const int32_t* mol_ptr, const int32_t* atom_mol, const int32_t* dst_ptr, const int32_t* src,
                          const int32_t* dst, const int32_t* perm, const int32_t* src_ptr, const int32_t* src_slot,
                          const int32_t* status_dev) {
  if (N <= 0 || E < 0 || B <= 0 || !batch || !mol_ptr || !atom_mol || !dst_ptr || !src_ptr || !status_dev)
    return LIGDIFF_ERR_ARG;
  if (E > 0 && (!edge_index || !src || !dst || !perm || !src_slot)) return LIGDIFF_ERR_ARG;
  if (N >= (1LL << 30) || E >= (1LL << 30)) return LIGDIFF_ERR_ARG;
  return LIGDIFF_OK;
}

extern "C" int ligdiff_pack_csr(const int64_t* edge_index, int64_t E, const int64_t* batch, int64_t N, int32_t B,
                                int32_t* mol_ptr, int32_t* atom_mol, int32_t* dst_ptr, int32_t* src, int32_t* dst,
                                int32_t* perm, int32_t* src_ptr, int32_t* src_slot, int32_t* status_dev,
                                void* workspace, size_t workspace_bytes, void* stream) {
  const int rc = pack_csr_check(edge_index, E, batch, N, B, mol_ptr, atom_mol, dst_ptr, src, dst, perm, src_ptr,
                                src_slot, status_dev);
  if (rc) return rc;
  if (!workspace) return LIGDIFF_ERR_ARG;
  if (workspace_bytes < ligdiff_pack_workspace_bytes(N, E)) return LIGDIFF_ERR_WORKSPACE;
  return pack_csr_run(edge_index, E, batch, N, B, mol_ptr, atom_mol, dst_ptr, src, dst, perm, src_ptr, src_slot,
                      status_dev, (int32_t*)workspace, (hipStream_t)stream);
}

// ------------------------------------------------------------------ packing without a workspace argument
// The counters of ligdiff_pack_csr live in a file-scope device array instead of a caller workspace, so
// the call allocates nothing and needs nothing allocated for it.  Every GPU the library runs on holds
// its own copy of the array (4.0 MiB) at its own address until the process ends; in host code the
// array's name is only a handle for hipGetSymbolAddress, whose result (per device) is what the kernels
// receive.  All calls on one GPU share that GPU's array: calls that overlap in time on one GPU race.
#define PACK_STATIC_MAX_ATOMS LIGDIFF_PACK_STATIC_MAX_ATOMS
static __device__ __align__(256) int32_t scratch[4 * (PACK_STATIC_MAX_ATOMS + 1)];

#include <mutex>
#define PACK_MAX_DEVICES 64
static std::mutex g_scratch_mutex;
static int32_t* g_scratch_ptr[PACK_MAX_DEVICES];   // device address of `scratch` on each device, once resolved

// The array is resolved for the CURRENT device; a stream of another device would hand its kernels the wrong
// device's address, so such a call is refused.
static int scratch_of_current_device(int32_t** out, hipStream_t stream) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= PACK_MAX_DEVICES) return LIGDIFF_ERR_LAUNCH;
  if (stream != nullptr) {
    hipDevice_t sdev = -1;
    if (hipStreamGetDevice(stream, &sdev) != hipSuccess) return LIGDIFF_ERR_LAUNCH;
    if ((int)sdev != dev) return LIGDIFF_ERR_ARG;
  }
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (g_scratch_ptr[dev] == nullptr) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)) != hipSuccess || p == nullptr) return LIGDIFF_ERR_LAUNCH;
    g_scratch_ptr[dev] = (int32_t*)p;
  }
  *out = g_scratch_ptr[dev];
  return LIGDIFF_OK;
}

extern "C" int ligdiff_pack_csr_static(const int64_t* edge_index, int64_t E, const int64_t* batch, int64_t N,
                                       int32_t B, int32_t* mol_ptr, int32_t* atom_mol, int32_t* dst_ptr, int32_t* src,
                                       int32_t* dst, int32_t* perm, int32_t* src_ptr, int32_t* src_slot,
                                       int32_t* status_dev, void* stream) {
  int rc = pack_csr_check(edge_index, E, batch, N, B, mol_ptr, atom_mol, dst_ptr, src, dst, perm, src_ptr, src_slot,
                          status_dev);
  if (rc) return rc;
  if (N > PACK_STATIC_MAX_ATOMS) return LIGDIFF_ERR_WORKSPACE;   // beyond the static array: use ligdiff_pack_csr
  int32_t* counters = nullptr;
  rc = scratch_of_current_device(&counters, (hipStream_t)stream);
  if (rc) return rc;
  // pack_csr_run zeroes the 4 (N + 1) counters it uses before counting, so what an earlier call left
  // behind (also beyond this call's N) never enters a result
  return pack_csr_run(edge_index, E, batch, N, B, mol_ptr, atom_mol, dst_ptr, src, dst, perm, src_ptr, src_slot,
                      status_dev, counters, (hipStream_t)stream);
}

// ------------------------------------------------------------------ template connectivity
// Device form of the generator's deterministic bond template (generation.py:562-668) for a
// whole batch of molecule sizes; one thread per bond, closed form in the bond index k:
//   n <= 5        chain            bond k = (k, k+1)
//   6 <= n <= 15  ring + subst.    k < 6: (k, (k+1)%6);  else s = k-6: ((2s)%6, 6+s)
//   n >= 16       fused rings      k < 6 ring; k < 11: (6+i, 6+(i+1)%5), i = k-6;
//                                  k = 11: (1,6); k = 12: (2,10); else s = k-13: (s%11, 11+s)
// Bond k of a molecule whose first atom is a0 and first directed edge is e0 writes
// (a,b) to edge e0+2k and (b,a) to e0+2k+1.  kind[e] = 1 for ring/fusion (aromatic) bonds.
__host__ __device__ inline int template_bonds(int n) {
  if (n <= 1) return 0;
  if (n <= 5) return n - 1;
  if (n <= 15) return n;
  return n + 2;
}

__global__ void k_template_edges(const int32_t* __restrict__ mol_ptr, const int64_t* __restrict__ edge_ptr, int32_t B,
                                 int32_t max_bonds, int64_t E, int64_t* __restrict__ edge_index,
                                 uint8_t* __restrict__ kind) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int mol = (int)(gid / max_bonds), k = (int)(gid % max_bonds);
  if (mol >= B) return;
  const int a0 = mol_ptr[mol], n = mol_ptr[mol + 1] - a0;
  if (k >= template_bonds(n)) return;
  int a, b, arom;
  if (n <= 5) { a = k; b = k + 1; arom = 0; }
  else if (k < 6) { a = k; b = (k + 1) % 6; arom = 1; }
  else if (n <= 15) { const int s = k - 6; a = (2 * s) % 6; b = 6 + s; arom = 0; }
  else if (k < 11) { const int i = k - 6; a = 6 + i; b = 6 + (i + 1) % 5; arom = 1; }
  else if (k == 11) { a = 1; b = 6; arom = 1; }
  else if (k == 12) { a = 2; b = 10; arom = 1; }
  else { const int s = k - 13; a = s % 11; b = 11 + s; arom = 0; }
  const int64_t e = edge_ptr[mol] + 2 * (int64_t)k;
  edge_index[e] = a0 + a;     edge_index[E + e] = a0 + b;
  edge_index[e + 1] = a0 + b; edge_index[E + e + 1] = a0 + a;
  if (kind != nullptr) { kind[e] = (uint8_t)arom; kind[e + 1] = (uint8_t)arom; }
}

extern "C" int64_t ligdiff_template_edge_count(int32_t n_atoms) { return 2 * (int64_t)template_bonds(n_atoms); }

extern "C" int ligdiff_template_edges(const int32_t* mol_ptr, const int64_t* edge_ptr, int32_t n_mols,
                                      int32_t max_mol_atoms, int64_t n_edges, int64_t* edge_index, uint8_t* kind,
                                      void* stream) {
  if (!mol_ptr || !edge_ptr || n_mols <= 0 || max_mol_atoms <= 0 || n_edges < 0) return LIGDIFF_ERR_ARG;
  if (n_edges == 0) return LIGDIFF_OK;
  if (!edge_index) return LIGDIFF_ERR_ARG;
  const int mb = template_bonds(max_mol_atoms) > 0 ? template_bonds(max_mol_atoms) : 1;
  hipLaunchKernelGGL(k_template_edges, dim3(ceil_div((int64_t)n_mols * mb, 256)), dim3(256), 0, (hipStream_t)stream,
                     mol_ptr, edge_ptr, n_mols, mb, n_edges, edge_index, kind);
  LD_CHECK_LAUNCH();
  return LIGDIFF_OK;
}

// ------------------------------------------------------------------ fully connected molecules
// Every ordered pair (i, j != i) of each molecule as a directed edge i -> j, pairs of a molecule in
// row-major (i, j) order: edge k of a molecule with n atoms has i = k / (n - 1) and j = k % (n - 1),
// moved past the diagonal.  One thread per edge; the molecule is found by bisection of edge_ptr
// (exclusive prefix of n (n - 1)).
__global__ void k_full_edges(const int32_t* __restrict__ mol_ptr, const int64_t* __restrict__ edge_ptr, int32_t B,
                             int64_t E, int64_t* __restrict__ edge_index) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  int lo = 0, hi = B;            // invariant: edge_ptr[lo] <= e < edge_ptr[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (edge_ptr[mid] <= e) lo = mid; else hi = mid;
  }
  const int a0 = mol_ptr[lo], n = mol_ptr[lo + 1] - a0;
  const int k = (int)(e - edge_ptr[lo]);
  const int i = k / (n - 1);
  int j = k % (n - 1);
  j += (j >= i) ? 1 : 0;
  edge_index[e] = a0 + i;
  edge_index[E + e] = a0 + j;
}

extern "C" int ligdiff_full_edges(const int32_t* mol_ptr, const int64_t* edge_ptr, int32_t n_mols, int64_t n_edges,
                                  int64_t* edge_index, void* stream) {
  if (!mol_ptr || !edge_ptr || n_mols <= 0 || n_edges < 0) return LIGDIFF_ERR_ARG;
  if (n_edges == 0) return LIGDIFF_OK;
  if (!edge_index) return LIGDIFF_ERR_ARG;
  hipLaunchKernelGGL(k_full_edges, dim3(ceil_div(n_edges, 256)), dim3(256), 0, (hipStream_t)stream, mol_ptr, edge_ptr,
                     n_mols, n_edges, edge_index);
  LD_CHECK_LAUNCH();
  return LIGDIFF_OK;
}
