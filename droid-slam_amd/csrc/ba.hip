// Dense bundle adjustment on the device: Gauss-Newton over SE(3) poses and per-pixel inverse depths
// with a Schur complement onto the poses.  Replaces ba_cuda and its kernels
// (reference src/droid_kernels.cu:185-433 projective_transform_kernel, :863-1007 accum, :1010-1124
// EEt6x6/Ev6x1/EvT6x1, :1126-1228 SparseBlock (Eigen, CPU, fp64), :1231-1320 schur_block,
// :1323-1443 ba_cuda).  Index conventions: SURVEY.md Appendix A.
//
// MI355X design (no host round trip anywhere; the reference crosses device->host->device >= 8x/iter):
//   prep      one block: unique source frames (depth blocks), CSR of edges by source frame -- built on
//             the device from ii (the reference argsorts on the CPU in every accum_cuda call);
//   build     FRAME-centric: workgroup = (source frame k, pixel strip).  Loops over k's out-edges with
//             lane = pixel, so per-frame sums (C_k, w_k, E_i[k]) are plain register accumulations in a
//             fixed edge order -- no segment-sum kernels, no atomics.  Uses linearity Ji = -Adj^T Jj:
//             only Hjj (21 entries) + vj are reduced per edge (seven 4-way wave reductions on v_permlane32/16_swap +
//             DPP, one partial per wave), the 6x6 adjoint is applied once per edge in fp64 (pose_blocks) and once
//             per pixel for E_i;
//   gram      Schur blocks of one depth block as a TRUE contraction over pixels: G = M^T diag(Q) M with
//             M = [E_slot0 .. E_slotn | w]  (HW x (6(n+1)+1)), on the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32), operands
//             straight from global memory (16-byte loads, no LDS staging), pixel groups interleaved over the 4 waves,
//             reduced in LDS, accumulated into the fp64 reduced camera system;
//   solve     hand-written blocked right-looking Cholesky in fp64 (64x64 blocks, one look-ahead launch per block
//             column, the panel in MFMA accumulator registers), the rhs is
//             carried as an extra block row so forward substitution is free; damping and the
//             "failure -> zero update" semantics of SparseBlock::solve are preserved;
//   backsub   dz = Q (w - sum_slots E^T dx) incl. the reference's row-skip quirk (EvT6x1_kernel :1114),
//             disps updated in place; poses retracted by exp(dx) * T.
// The reduced camera system is accumulated with fp64 atomics (order-dependent only at the 1e-16
// level); everything else is deterministic.
#include "common.h"
#include <algorithm>
#include <type_traits>

namespace {

using namespace dh;

constexpr int NB = 64;            // Cholesky block
constexpr int PPT = 3;            // pixels per thread in build/backsub strips
constexpr int STRIP = 256 * PPT;  // pixels per strip
constexpr int HP_STRIDE = 28;     // per (edge, strip, wave) partial: 21 Hjj + 6 vj (+1 pad)
constexpr int GS = 10;            // slots per Gram chunk: 10*6 + 1 (w) = 61 <= 64 MFMA columns
constexpr int GCOLS = 64;
constexpr float ALPHA_PRIOR = 0.05f;   // src/droid_kernels.cu:1405

constexpr int FLOW_HDR = 32;        // ints ahead of the ready flags in BaLayout::flow (ticket counter and watchdog word on their own line)

struct BaLayout {
  size_t kmap, kx, eoff, cursor, eidx, meta;          // int32 arrays
  size_t Q, W, Ei, Ej, Hpart;                          // f32
  size_t H, x, Linv, Ldiag;                            // f64
  size_t flow;                                         // int32: ticket counter, watchdog word, one ready flag per block of L (dataflow Cholesky)
  size_t dx;                                           // f32 [P,6]
  size_t total;
  int P, n, nbk, npad, ld, NS, Kmax;
};

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// compute units of the current device (grid of the persistent Cholesky); queried once per device
inline int device_cus() {
  static int cu_count[64];
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
    if (!cu_count[dev]) { int n = 0; if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cu_count[dev] = n; }
    if (cu_count[dev]) cus = cu_count[dev];
  }
  return cus;
}

BaLayout make_layout(int F, int E, int HW, int t0, int t1, int motion_only) {
  BaLayout L{};
  L.P = t1 - t0;
  L.n = 6 * L.P;
  L.nbk = (L.n + NB - 1) / NB;
  L.npad = L.nbk * NB;
  L.ld = L.npad;
  L.NS = (HW + STRIP - 1) / STRIP;
  L.Kmax = F;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
  L.kmap = take(sizeof(int) * (size_t)F);
  L.kx = take(sizeof(int) * (size_t)F);
  L.eoff = take(sizeof(int) * ((size_t)F + 1));
  L.cursor = take(sizeof(int) * (size_t)F);
  L.eidx = take(sizeof(int) * (size_t)(E > 0 ? E : 1));
  L.meta = take(sizeof(int) * 16);
  L.Hpart = take(sizeof(float) * (size_t)(E > 0 ? E : 1) * L.NS * 4 * HP_STRIDE);
  if (!motion_only) {
    L.Q = take(sizeof(float) * (size_t)F * HW);
    L.W = take(sizeof(float) * (size_t)F * HW);
    L.Ei = take(sizeof(float) * (size_t)F * 6 * HW);
    L.Ej = take(sizeof(float) * (size_t)(E > 0 ? E : 1) * 6 * HW);
  }
  L.H = take(sizeof(double) * (size_t)(L.npad + NB) * L.ld);
  L.x = take(sizeof(double) * (size_t)L.npad);
  L.Linv = take(sizeof(double) * (size_t)L.nbk * NB * NB);
  L.Ldiag = take(sizeof(double) * (size_t)L.nbk * NB * NB);
  L.dx = take(sizeof(float) * (size_t)(L.P > 0 ? L.P : 1) * 6);
  L.flow = take(sizeof(int) * (FLOW_HDR + (size_t)(L.nbk + 1) * (L.nbk > 0 ? L.nbk : 1)));
  L.total = off;
  return L;
}

// meta[0] = K (number of depth blocks), meta[1] = Cholesky failure flag of the CURRENT iteration (cleared by
// ba_damp_kernel), meta[2] = argument flags: bit 0 an edge with ii or jj outside [0, F) (such edges are dropped),
// bit 1 eta does not have one row per depth block.  Any bit of meta[2] turns the whole call into a no-op update.

// ------------------------------------------------------------------------------------------ prep
__global__ __launch_bounds__(1024) void ba_prep_kernel(
    const int64_t* __restrict__ ii, const int64_t* __restrict__ jj, int E, int F, int t0, int t1, int n_eta_rows,
    int* __restrict__ kmap, int* __restrict__ kx, int* __restrict__ eoff, int* __restrict__ cursor,
    int* __restrict__ eidx, int* __restrict__ meta) {
  __shared__ int s_part[1024];
  __shared__ int s_part2[1024];
  const int tid = threadIdx.x;
  // kmap doubles as "present" flag, cursor as degree
  for (int f = tid; f < F; f += 1024) { kmap[f] = (f >= t0 && f < t1) ? 1 : 0; cursor[f] = 0; }
  if (tid == 0) { meta[1] = 0; meta[2] = 0; }
  __syncthreads();
  for (int e = tid; e < E; e += 1024) {
    const long f = ii[e], j = jj[e];
    if (f >= 0 && f < F && j >= 0 && j < F) { kmap[f] = 1; atomicAdd(&cursor[f], 1); }
    else atomicOr(&meta[2], 1);                       // edge dropped: it never enters the CSR
  }
  __syncthreads();
  // block scan over frames: each thread owns a contiguous chunk
  const int chunk = (F + 1023) / 1024;
  const int f0 = tid * chunk, f1 = min(F, f0 + chunk);
  int np = 0, nd = 0;
  for (int f = f0; f < f1; ++f) { np += kmap[f]; nd += kmap[f] ? cursor[f] : 0; }
  s_part[tid] = np; s_part2[tid] = nd;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    int a = 0, b = 0;
    if (tid >= off) { a = s_part[tid - off]; b = s_part2[tid - off]; }
    __syncthreads();
    s_part[tid] += a; s_part2[tid] += b;
    __syncthreads();
  }
  int kbase = s_part[tid] - np, ebase = s_part2[tid] - nd;
  for (int f = f0; f < f1; ++f) {
    if (kmap[f]) {
      const int d = cursor[f];
      kx[kbase] = f; eoff[kbase] = ebase; kmap[f] = kbase; cursor[f] = 0;
      kbase++; ebase += d;
    } else {
      kmap[f] = -1;
    }
  }
  if (tid == 1023) { meta[0] = s_part[1023]; }
  __syncthreads();
  const int K = meta[0];
  if (tid == 0) {
    eoff[K] = s_part2[1023];
    if (n_eta_rows >= 0 && n_eta_rows != K) atomicOr(&meta[2], 2);      // the reference's broadcast would fail (:1407)
  }
  __syncthreads();
  for (int e = tid; e < E; e += 1024) {
    const long f = ii[e], j = jj[e];
    if (f >= 0 && f < F && j >= 0 && j < F) {
      const int k = kmap[f];
      const int slot = atomicAdd(&cursor[f], 1);
      eidx[eoff[k] + slot] = e;
    }
  }
  __syncthreads();
  // stable order inside each segment (ascending edge id): insertion sort, segments are short
  for (int k = tid; k < K; k += 1024) {
    const int a = eoff[k], b = eoff[k + 1];
    for (int i = a + 1; i < b; ++i) {
      int v = eidx[i], j = i - 1;
      while (j >= a && eidx[j] > v) { eidx[j + 1] = eidx[j]; --j; }
      eidx[j + 1] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------ build
struct EdgeGeom { SE3f T; bool stereo; };

__device__ __forceinline__ EdgeGeom edge_geom(const float* __restrict__ poses, int i, int j) {
  EdgeGeom g;
  g.stereo = (i == j);
  if (g.stereo) g.T = stereo_rel();
  else g.T = rel(load_pose(poses + 7 * (long)i), load_pose(poses + 7 * (long)j));
  return g;
}

// per-pixel residual / Jacobian terms of one edge (reference :290-385, restated)
struct PixTerms { float ru, rv, wu, wv, Jju[6], Jjv[6], Jzu, Jzv; };

__device__ __forceinline__ PixTerms pix_terms(const SE3f& T, float X, float Y, float h,
                                              float tu, float tv, float wwu, float wwv,
                                              float fx, float fy, float cx, float cy) {
  PixTerms o;
  Vec3 R = rot(T.q, {X, Y, 1.f});
  const float x = R.x + h * T.t.x, y = R.y + h * T.t.y, z = R.z + h * T.t.z;
  const bool bad = z < DH_MIN_DEPTH;
  const float d = bad ? 0.f : 1.f / z;
  const float d2 = d * d;
  o.wu = bad ? 0.f : 0.001f * wwu;
  o.wv = bad ? 0.f : 0.001f * wwv;
  o.ru = tu - (fx * d * x + cx);
  o.rv = tv - (fy * d * y + cy);
  o.Jju[0] = fx * (h * d); o.Jju[1] = 0.f; o.Jju[2] = fx * (-x * h * d2);
  o.Jju[3] = fx * (-x * y * d2); o.Jju[4] = fx * (1.f + x * x * d2); o.Jju[5] = fx * (-y * d);
  o.Jjv[0] = 0.f; o.Jjv[1] = fy * (h * d); o.Jjv[2] = fy * (-y * h * d2);
  o.Jjv[3] = fy * (-1.f - y * y * d2); o.Jjv[4] = fy * (x * y * d2); o.Jjv[5] = fy * (x * d);
  o.Jzu = fx * (T.t.x * d - T.t.z * (x * d2));
  o.Jzv = fy * (T.t.y * d - T.t.z * (y * d2));
  return o;
}

template <bool MOTION_ONLY>
__global__ __launch_bounds__(256) void ba_build_kernel(
    const float* __restrict__ poses, const float* __restrict__ disps, const float* __restrict__ intr,
    const float* __restrict__ disps_sens, const float* __restrict__ targets, const float* __restrict__ weights,
    const float* __restrict__ eta, int n_eta_rows, const float* __restrict__ alpha, const int64_t* __restrict__ jj,
    const int* __restrict__ kx, const int* __restrict__ eoff, const int* __restrict__ eidx,
    const int* __restrict__ meta, int HW, int wd, int NS,
    float* __restrict__ Q, float* __restrict__ W, float* __restrict__ Ei, float* __restrict__ Ej,
    float* __restrict__ Hpart) {
  const int k = blockIdx.x;
  if (k >= meta[0]) return;
  const int strip = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int f = kx[k];
  const float fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];

  int px[PPT]; bool ok[PPT]; float Xn[PPT], Yn[PPT], h[PPT];
  float Cs[PPT], Ws[PPT], Eacc[PPT][6];
#pragma unroll
  for (int q = 0; q < PPT; ++q) {
    px[q] = strip * STRIP + q * 256 + tid;
    ok[q] = px[q] < HW;
    const int p = ok[q] ? px[q] : 0;
    const int row = p / wd, col = p - row * wd;
    Xn[q] = ((float)col - cx) / fx;
    Yn[q] = ((float)row - cy) / fy;
    h[q] = disps[(long)f * HW + p];
    Cs[q] = 0.f; Ws[q] = 0.f;
#pragma unroll
    for (int n = 0; n < 6; ++n) Eacc[q][n] = 0.f;
  }

  const int e0 = eoff[k], e1 = eoff[k + 1];
  // targets / weights of the NEXT edge travel from HBM while this edge is being reduced
  float tw[PPT][4];
  auto fetch_tw = [&](int e, float (&o)[PPT][4]) {
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
      const int p = ok[q] ? px[q] : 0;
      o[q][0] = targets[((long)e * 2 + 0) * HW + p];
      o[q][1] = targets[((long)e * 2 + 1) * HW + p];
      o[q][2] = weights[((long)e * 2 + 0) * HW + p];
      o[q][3] = weights[((long)e * 2 + 1) * HW + p];
    }
  };
  int e = e0 < e1 ? eidx[e0] : 0;
  if (e0 < e1) fetch_tw(e, tw);
  for (int ei = e0; ei < e1; ++ei) {
    const int e_next = eidx[min(ei + 1, e1 - 1)];
    float ntw[PPT][4];
    fetch_tw(e_next, ntw);
    const int j = (int)jj[e];
    const EdgeGeom g = edge_geom(poses, f, j);
    float hj[21], vj[6];
#pragma unroll
    for (int l = 0; l < 21; ++l) hj[l] = 0.f;
#pragma unroll
    for (int n = 0; n < 6; ++n) vj[n] = 0.f;
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
      const int p = ok[q] ? px[q] : 0;
      const float tu = tw[q][0], tv = tw[q][1];
      const float wu0 = ok[q] ? tw[q][2] : 0.f;
      const float wv0 = ok[q] ? tw[q][3] : 0.f;
      PixTerms t = pix_terms(g.T, Xn[q], Yn[q], h[q], tu, tv, wu0, wv0, fx, fy, cx, cy);
      if (!MOTION_ONLY) {
        Cs[q] += t.wu * t.Jzu * t.Jzu + t.wv * t.Jzv * t.Jzv;
        Ws[q] += t.wu * t.ru * t.Jzu + t.wv * t.rv * t.Jzv;
      }
      const float wu = g.stereo ? 0.f : t.wu, wv = g.stereo ? 0.f : t.wv;   // pose terms off for stereo pairs
      int l = 0;
#pragma unroll
      for (int a = 0; a < 6; ++a) {
#pragma unroll
        for (int b = 0; b <= a; ++b) { hj[l] += wu * t.Jju[a] * t.Jju[b] + wv * t.Jjv[a] * t.Jjv[b]; ++l; }
        vj[a] += wu * t.ru * t.Jju[a] + wv * t.rv * t.Jjv[a];
      }
      if (!MOTION_ONLY) {
        float ej[6], ei6[6];
#pragma unroll
        for (int a = 0; a < 6; ++a) ej[a] = wu * t.Jzu * t.Jju[a] + wv * t.Jzv * t.Jjv[a];
        adjT(g.T, ej, ei6);
#pragma unroll
        for (int a = 0; a < 6; ++a) {
          Eacc[q][a] -= ei6[a];
          if (ok[q]) Ej[((long)e * 6 + a) * HW + p] = ej[a];
        }
      }
    }
    // block reduction of the 27 per-edge sums: seven 4-way wave reductions (rows of the result register hold the
    // totals of values 4i, 4i+2, 4i+1, 4i+3), then the 4 waves through LDS
    float r4[7];
#pragma unroll
    for (int i = 0; i < 5; ++i) r4[i] = wave_sum4(hj[4 * i], hj[4 * i + 1], hj[4 * i + 2], hj[4 * i + 3]);
    r4[5] = wave_sum4(hj[20], vj[0], vj[1], vj[2]);
    r4[6] = wave_sum4(vj[3], vj[4], vj[5], 0.f);
    // every wave leaves its own partial (ba_pose_blocks_kernel sums the NS * 4 of an edge in a fixed order): no LDS, no
    // barrier -- the waves of a workgroup never wait for each other
    if ((lane & 15) == 0) {
      const int row = lane >> 4, sub = ((row & 1) << 1) | (row >> 1);       // row 0,1,2,3 -> value 0,2,1,3 of the group
      float* hp = Hpart + (((long)e * NS + strip) * 4 + wave) * HP_STRIDE;
#pragma unroll
      for (int i = 0; i < 7; ++i) hp[4 * i + sub] = r4[i];
    }
    e = e_next;
#pragma unroll
    for (int q = 0; q < PPT; ++q)
#pragma unroll
      for (int c = 0; c < 4; ++c) tw[q][c] = ntw[q][c];
  }

  if (!MOTION_ONLY) {
    const bool eta_ok = k < n_eta_rows;                  // a row-count mismatch is flagged by prep (meta[2] bit 1)
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
      if (!ok[q]) continue;
      const int p = px[q];
      const float ds = disps_sens[(long)f * HW + p];
      // depth prior: the reference's constant alpha = 0.05 (:1405), or per pixel (dh_ba_ex: confidence of the sensor depth).
      // Zero (or negative) confidence = no prior at that pixel: it is damped by eta like a pixel without sensor depth
      // (alpha = 0 with all weights zero would make C = 0)
      const float al = alpha ? alpha[(long)f * HW + p] : ALPHA_PRIOR;
      const bool m = ds > 0.f && al > 0.f;
      const float C = Cs[q] + (m ? al : (eta_ok ? eta[(long)k * HW + p] : 1.f));
      const float w = Ws[q] - (m ? al * (h[q] - ds) : 0.f);
      Q[(long)k * HW + p] = 1.f / C;
      W[(long)k * HW + p] = w;
#pragma unroll
      for (int a = 0; a < 6; ++a) Ei[((long)k * 6 + a) * HW + p] = Eacc[q][a];
    }
  }
}

// ---- per-edge pose Hessian blocks from the reduced Hjj / vj via the adjoint ---------------------
__global__ __launch_bounds__(64) void ba_pose_blocks_kernel(
    const float* __restrict__ poses, const int64_t* __restrict__ ii, const int64_t* __restrict__ jj,
    const float* __restrict__ Hpart, int E, int F, int NS, int t0, int P, double* __restrict__ H, int ld, int brow) {
  const int e = blockIdx.x;
  const int tid = threadIdx.x;
  __shared__ double s_h[6][6], s_v[6], s_A[6][6], s_AH[6][6];
  if (ii[e] < 0 || ii[e] >= F || jj[e] < 0 || jj[e] >= F) return;   // dropped edge (flagged by prep): its Hpart rows were never written
  const int i = (int)ii[e], j = (int)jj[e];
  if (i == j) return;                                    // stereo pair: no pose terms
  const int pi = i - t0, pj = j - t0;
  const bool vi_ok = pi >= 0 && pi < P, vj_ok = pj >= 0 && pj < P;
  if (!vi_ok && !vj_ok) return;
  if (tid < 27) {
    double s = 0;
    for (int st = 0; st < NS * 4; ++st) s += (double)Hpart[((long)e * NS * 4 + st) * HP_STRIDE + tid];       // strips x waves
    if (tid < 21) {
      int a = 0, rem = tid;
      while (rem > a) { rem -= a + 1; ++a; }
      s_h[a][rem] = s; s_h[rem][a] = s;
    } else {
      s_v[tid - 21] = s;
    }
  }
  if (tid >= 32 && tid < 38) {                           // column c of A = -Adj(Tij)^T
    const int c = tid - 32;
    EdgeGeom g = edge_geom(poses, i, j);
    float ec[6] = {0, 0, 0, 0, 0, 0}, col[6];
    ec[c] = 1.f;
    adjT(g.T, ec, col);
    for (int r = 0; r < 6; ++r) s_A[r][c] = -(double)col[r];
  }
  __syncthreads();
  if (tid < 36) {
    const int r = tid / 6, c = tid % 6;
    double s = 0;
    for (int m = 0; m < 6; ++m) s += s_A[r][m] * s_h[m][c];
    s_AH[r][c] = s;                                      // Hij = A Hjj
  }
  __syncthreads();
  if (tid < 36) {
    const int r = tid / 6, c = tid % 6;
    // (lower triangle only, like the Schur blocks)
    if (vj_ok && r >= c) atomicAdd(&H[(long)(6 * pj + r) * ld + 6 * pj + c], s_h[r][c]);
    if (vi_ok && vj_ok) {
      if (pi > pj) atomicAdd(&H[(long)(6 * pi + r) * ld + 6 * pj + c], s_AH[r][c]);     // Hij
      else atomicAdd(&H[(long)(6 * pj + r) * ld + 6 * pi + c], s_AH[c][r]);             // Hji = Hij^T
    }
    if (vi_ok && r >= c) {
      double s = 0;
      for (int m = 0; m < 6; ++m) s += s_AH[r][m] * s_A[c][m];              // Hii = A Hjj A^T
      atomicAdd(&H[(long)(6 * pi + r) * ld + 6 * pi + c], s);
    }
  } else if (tid < 42) {
    const int r = tid - 36;
    if (vj_ok) atomicAdd(&H[(long)brow * ld + 6 * pj + r], s_v[r]);
    if (vi_ok) {
      double s = 0;
      for (int m = 0; m < 6; ++m) s += s_A[r][m] * s_v[m];
      atomicAdd(&H[(long)brow * ld + 6 * pi + r], s);
    }
  }
}

// ---- Schur complement blocks of one depth block: Gram matrix on the fp32 MFMA -------------------
// G = M^T diag(Q) M over the pixels of depth block k, M = [E_i | E_ij of the block's edges | w] (61 of 64 columns per
// chunk of GS slots).  The pixel sum is order-free, so the MFMA operands come STRAIGHT from global memory: lane
// (c = lane & 15, g = lane >> 4) reads four consecutive pixels of column 16 t + c as one 16-byte load and feeds element e
// to k-step e -- A and B see the same pixel for the same (k-step, lane group), which is all the contraction needs.  No LDS
// staging, no barrier in the pixel loop; the waves of a workgroup take interleaved 16-pixel groups and meet once, in the
// cross-wave reduction of the finished 64x64 Gram.  Q scales the A side only.
using f32x4 = __attribute__((ext_vector_type(4))) float;

struct GramCols { const float* p[4]; unsigned valid; };      // lane's column of each 16-column tile; bit t = column exists

// the column pointers travel through LDS, which costs them their address space: without the cast the loads become FLAT
// loads, which the compiler has to drain with vmcnt(0) before anything else -- no prefetch would survive that
typedef const float __attribute__((address_space(1)))* gram_gptr;
typedef const f32x4 __attribute__((address_space(1)))* gram_gptr4;    // (a native vector: HIP's float4 class would copy through a generic reference)

template <bool VEC4>
__device__ __forceinline__ f32x4 gram_ld(const float* col, int pix, int p_end) {
  if (VEC4) return *(gram_gptr4)(col + pix);
  gram_gptr c = (gram_gptr)col;
  return f32x4{c[pix], c[min(pix + 1, p_end - 1)], c[min(pix + 2, p_end - 1)], c[min(pix + 3, p_end - 1)]};
}

// One 16-pixel group of a wave: NA column tiles on the A side (first tile TA0), NBT on the B side (0: B = A, a chunk
// against itself) and Q.
template <int NA, int NBT> struct GramTile { f32x4 a[NA], b[NBT ? NBT : 1], q; };

template <bool VEC4, int NA, int NBT, int TA0>
__device__ __forceinline__ void gram_fetch(const GramCols& CA, const GramCols& CB, const float* __restrict__ Qk, int p0, int g,
                                           int p_end, GramTile<NA, NBT>& T) {
  const int pix = min(p0 + 4 * g, p_end - 1) & (VEC4 ? ~3 : ~0);       // clamped: every lane loads from inside the strip
#pragma unroll
  for (int t = 0; t < NA; ++t) T.a[t] = gram_ld<VEC4>(CA.p[TA0 + t], pix, p_end);
#pragma unroll
  for (int t = 0; t < NBT; ++t) T.b[t] = gram_ld<VEC4>(CB.p[t], pix, p_end);
  T.q = gram_ld<VEC4>(Qk, pix, p_end);
}

// acc[ti * 4 + tj] += (Q A_ti)^T B_tj; pixels past the end of the strip (ragged last group) and columns that do not exist
// contribute 0.  NBT == 0: upper tiles of the symmetric product only.
template <int NA, int NBT, int TA0>
__device__ __forceinline__ void gram_mfma(const GramCols& CA, const GramCols& CB, const GramTile<NA, NBT>& T, int pix, int p_end,
                                          f32x4 (&acc)[NA * 4]) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float qe = (pix + e < p_end) ? T.q[e] : 0.f;
    float a[NA], bb[4];
#pragma unroll
    for (int t = 0; t < NA; ++t) {
      const float x = ((CA.valid >> (TA0 + t)) & 1) ? T.a[t][e] : 0.f;
      a[t] = x * qe;
      if (NBT == 0) bb[t] = x;
    }
#pragma unroll
    for (int t = 0; t < NBT; ++t) bb[t] = ((CB.valid >> t) & 1) ? T.b[t][e] : 0.f;
#pragma unroll
    for (int ti = 0; ti < NA; ++ti)
#pragma unroll
      for (int tj = 0; tj < (NBT ? NBT : NA); ++tj)
        if (NBT != 0 || tj >= ti)
          acc[ti * 4 + tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ti], bb[tj], acc[ti * 4 + tj], 0, 0, 0);
  }
}

// two groups in flight, ping-pong (no register copies), every fetch unconditional -- addresses are clamped into the strip
// and a group past its end has all its pixels masked -- so that the compiler can wait for exactly the older group's loads
template <bool VEC4, int NA, int NBT, int TA0>
__device__ __forceinline__ void gram_accumulate(const GramCols& CA, const GramCols& CB, const float* __restrict__ Qk,
                                                int p_begin, int p_end, int wave, int g, f32x4 (&acc)[NA * 4]) {
  GramTile<NA, NBT> T0, T1;
  int p0 = p_begin + 16 * wave;
  gram_fetch<VEC4, NA, NBT, TA0>(CA, CB, Qk, p0, g, p_end, T0);
  for (; p0 < p_end; p0 += 128) {
    gram_fetch<VEC4, NA, NBT, TA0>(CA, CB, Qk, p0 + 64, g, p_end, T1);
    __builtin_amdgcn_sched_barrier(0);                   // the loads stay ahead of the other group's MFMAs
    gram_mfma<NA, NBT, TA0>(CA, CB, T0, p0 + 4 * g, p_end, acc);
    __builtin_amdgcn_sched_barrier(0);
    gram_fetch<VEC4, NA, NBT, TA0>(CA, CB, Qk, p0 + 128, g, p_end, T0);
    __builtin_amdgcn_sched_barrier(0);
    gram_mfma<NA, NBT, TA0>(CA, CB, T1, p0 + 64 + 4 * g, p_end, acc);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// the waves' partial Grams (tiles ti0 .. ti0 + NA - 1 of the rows) summed into s_out in a fixed order
template <int NA, int NTJ>
__device__ __forceinline__ void gram_reduce(float* __restrict__ s_out, const f32x4 (&acc)[NA * 4], int ti0, bool upper_only,
                                            int wave, int lane) {
  for (int w = 0; w < 4; ++w) {
    __syncthreads();
    if (wave == w) {
#pragma unroll
      for (int ti = 0; ti < NA; ++ti)
#pragma unroll
        for (int tj = 0; tj < NTJ; ++tj) {
          if (upper_only && tj < ti0 + ti) continue;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = (ti0 + ti) * 16 + (lane >> 4) * 4 + r, col = tj * 16 + (lane & 15);
            if (w == 0) s_out[row * GCOLS + col] = acc[ti * 4 + tj][r];
            else s_out[row * GCOLS + col] += acc[ti * 4 + tj][r];
          }
        }
    }
  }
}

// DIAG: the products of every chunk with itself (all there is for a depth block with at most GS - 1 edges); !DIAG: the
// chunk pairs of the larger blocks.  Two kernels so that the common one is compiled for its own, smaller register budget
// (4 waves per SIMD: the pixel loop lives on memory-level parallelism).
template <bool DIAG>
__global__ __launch_bounds__(256, DIAG ? 4 : 2) void ba_gram_kernel(
    const float* __restrict__ Q, const float* __restrict__ W, const float* __restrict__ Ei,
    const float* __restrict__ Ej, const int64_t* __restrict__ jj, const int* __restrict__ kx,
    const int* __restrict__ eoff, const int* __restrict__ eidx, const int* __restrict__ meta,
    int HW, int NSG, int t0, int P, double* __restrict__ H, int ld, int brow) {
  const int k = blockIdx.x;
  if (k >= meta[0]) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // scalar: the pixel loop below branches uniformly
  const int f = kx[k];
  const int e0 = eoff[k];
  const int nslots = 1 + (eoff[k + 1] - e0);
  const int nchunks = (nslots + GS - 1) / GS;
  const int per = ((HW + NSG - 1) / NSG + 63) / 64 * 64;
  const int p_begin = blockIdx.y * per, p_end = min(HW, p_begin + per);
  if (p_begin >= p_end) return;
  const bool vec4 = (HW & 3) == 0;              // 16-byte loads need every column (a multiple of HW floats) aligned

  __shared__ float s_out[GCOLS * GCOLS];        // cross-wave reduction
  __shared__ int s_pose[2][GS];
  __shared__ const float* s_col[2][GCOLS];      // per-column source rows (nullptr = zero column)
  const float* Qk = Q + (long)k * HW;

  if (!DIAG && nchunks < 2) return;
  const int g = lane >> 4;
  for (int ca = 0; ca < nchunks; ++ca) {
    for (int cb = DIAG ? ca : ca + 1; cb < (DIAG ? ca + 1 : nchunks); ++cb) {
      constexpr bool same = DIAG;
      // slots of the two chunks; the w column (DIAG only) sits right behind the last real column, so a short chunk is a
      // short product: only the 16-column tiles that hold something are multiplied
      const int nsa = min(GS, nslots - ca * GS), nsb = min(GS, nslots - cb * GS);
      const int nta = (6 * nsa + (DIAG ? 1 : 0) + 15) / 16, ntb = (6 * nsb + 15) / 16;
      __syncthreads();
      if (tid < 2 * GCOLS) {
        const int which = tid / GCOLS, col = tid % GCOLS, c = which ? cb : ca, ns = which ? nsb : nsa;
        const float* ptr = nullptr;
        if (col < 6 * ns) {
          const int sl = c * GS + col / 6, a = col % 6;
          ptr = (sl == 0) ? Ei + ((long)k * 6 + a) * HW : Ej + ((long)eidx[e0 + sl - 1] * 6 + a) * HW;
        } else if (DIAG && col == 6 * ns) {
          ptr = W + (long)k * HW;
        }
        s_col[which][col] = ptr;
      }
      if (tid < 2 * GS) {
        const int which = tid / GS, sl = (which ? cb : ca) * GS + tid % GS;
        int pose = -1;
        if (sl < nslots) {
          const int fr = (sl == 0) ? f : (int)jj[eidx[e0 + sl - 1]];
          const int r = fr - t0;
          if (r >= 0 && r < P) pose = r;
        }
        s_pose[which][tid % GS] = pose;
      }
      __syncthreads();

      GramCols CA, CB;
      CA.valid = CB.valid = 0;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float* pa = s_col[0][t * 16 + (lane & 15)];
        const float* pb = s_col[same ? 0 : 1][t * 16 + (lane & 15)];
        CA.valid |= (pa != nullptr) << t; CB.valid |= (pb != nullptr) << t;
        CA.p[t] = pa ? pa : Qk;                 // a column that does not exist reads Q (in bounds) and is masked to 0
        CB.p[t] = pb ? pb : Qk;
      }
      if (same) {
        auto diag = [&](auto nt_c) {
          constexpr int NT = decltype(nt_c)::value;
          f32x4 acc[NT * 4];
#pragma unroll
          for (int t = 0; t < NT * 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
          gram_accumulate<true, NT, 0, 0>(CA, CB, Qk, p_begin, p_end, wave, g, acc);
          gram_reduce<NT, NT>(s_out, acc, 0, true, wave, lane);
        };
        if (!vec4) {
          f32x4 acc[16];
#pragma unroll
          for (int t = 0; t < 16; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
          gram_accumulate<false, 4, 0, 0>(CA, CB, Qk, p_begin, p_end, wave, g, acc);
          gram_reduce<4, 4>(s_out, acc, 0, true, wave, lane);
        } else if (nta == 1) diag(std::integral_constant<int, 1>{});
        else if (nta == 2) diag(std::integral_constant<int, 2>{});
        else if (nta == 3) diag(std::integral_constant<int, 3>{});
        else diag(std::integral_constant<int, 4>{});
      } else {
        // two chunks against each other (a depth block with more than GS - 1 edges): passes of two A tiles keep the register
        // budget low; the second pass shifts A tiles 2, 3 into the places of 0, 1
        for (int pass = 0; pass < 2 && 2 * pass < nta; ++pass) {
          if (pass == 1) { CA.p[0] = CA.p[2]; CA.p[1] = CA.p[3]; CA.valid >>= 2; }
          const int na = min(2, nta - 2 * pass);
          auto off = [&](auto na_c, auto nb_c) {
            constexpr int NA = decltype(na_c)::value, NBT = decltype(nb_c)::value;
            f32x4 acc[NA * 4];
#pragma unroll
            for (int t = 0; t < NA * 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            gram_accumulate<true, NA, NBT, 0>(CA, CB, Qk, p_begin, p_end, wave, g, acc);
            gram_reduce<NA, NBT>(s_out, acc, 2 * pass, false, wave, lane);
          };
          using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
          using I3 = std::integral_constant<int, 3>; using I4 = std::integral_constant<int, 4>;
          if (!vec4) {
            f32x4 acc[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            gram_accumulate<false, 2, 4, 0>(CA, CB, Qk, p_begin, p_end, wave, g, acc);
            gram_reduce<2, 4>(s_out, acc, 2 * pass, false, wave, lane);
          } else if (na == 1) {
            if (ntb == 1) off(I1{}, I1{}); else if (ntb == 2) off(I1{}, I2{}); else if (ntb == 3) off(I1{}, I3{}); else off(I1{}, I4{});
          } else {
            if (ntb == 1) off(I2{}, I1{}); else if (ntb == 2) off(I2{}, I2{}); else if (ntb == 3) off(I2{}, I3{}); else off(I2{}, I4{});
          }
        }
      }
      __syncthreads();
      // G[row=(slot a, r)][col=(slot b, c)] -> S block (pose_a, pose_b); column 6 nsa = E Q w
      for (int o = tid; o < 6 * nsa * 6 * (same ? nsa : nsb); o += 256) {
        const int ncol = 6 * (same ? nsa : nsb);
        const int row = o / ncol, col = o % ncol;
        const int pa = s_pose[0][row / 6], pbq = s_pose[same ? 0 : 1][col / 6];
        if (pa < 0 || pbq < 0) continue;
        const bool lower = same && (col >> 4) < (row >> 4);
        const double v = (double)(lower ? s_out[col * GCOLS + row] : s_out[row * GCOLS + col]);
        // only the lower triangle of the reduced camera system is ever read (Cholesky, packed exchange): half the atomics
        const int R = 6 * pa + row % 6, C = 6 * pbq + col % 6;
        if (R >= C) atomicAdd(&H[(long)R * ld + C], -v);
        if (!same && C >= R) atomicAdd(&H[(long)C * ld + R], -v);
      }
      if (same) {
        for (int o = tid; o < 6 * nsa; o += 256) {
          const int pa = s_pose[0][o / 6];
          if (pa >= 0) atomicAdd(&H[(long)brow * ld + 6 * pa + o % 6], -(double)s_out[o * GCOLS + 6 * nsa]);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------ solve
// (also clears the Cholesky failure flag: SparseBlock::solve judges every iteration on its own, droid_kernels.cu:1201-1221)
__global__ void ba_damp_kernel(double* __restrict__ H, int ld, int n, int npad, double lm, double ep, int* __restrict__ meta,
                               int* __restrict__ flow, int nflow) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) meta[1] = 0;
  for (int k = i; k < nflow; k += gridDim.x * blockDim.x) flow[k] = 0;      // ticket counter and ready flags of chol_flow_kernel
  if (i >= npad) return;
  if (i < n) { const double d = H[(long)i * ld + i]; H[(long)i * ld + i] = d + ep + lm * d; }
  else H[(long)i * ld + i] = 1.0;
}

// ---- blocked Cholesky, NB = 64 ------------------------------------------------------------------------
// Default schedule: ONE launch per block column (chol_step_kernel below: trailing update of column j fused with the
// factorisation of column j+1; 5.79 -> 5.29 ms per global BA at 512 keyframes); DH_CHOL_LOOKAHEAD=0 selects the
// two-launch schedule described here, which is also what step 0 and the parity tests of both schedules use:
// step j:  chol_panel_kernel  one workgroup per block row r > j (incl. the rhs row): EVERY workgroup factors the
//                             diagonal block A_jj itself in LDS (redundant work is free, a separate launch and
//                             its ~2 us dependency gap are not), inverts the factor (W = L_jj^-1) and forms
//                             L_rj = A_rj W^T on the fp64 MFMA; workgroup 0 also keeps W_j for the back solve;
//          chol_update_kernel A_rc -= L_rj L_cj^T for j < c <= r on the fp64 MFMA (v_mfma_f64_16x16x4_f64).
// Inside a 64x64 block the factorisation is blocked again by 16: the 16x16 diagonal sub-block is factorised
// by ONE wave with row r in the registers of lane r (cross-lane traffic = v_readlane broadcasts, no barriers),
// the panel below it by row-wise substitution, the trailing update by all 4 waves.
#ifdef DH_CHOL_TS                  // scripts/ubench/chol_ts.hip: phase timestamps of workgroup 0 (never defined in the library build)
__device__ unsigned long long g_chol_ts[128];
#define DH_TS(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) { g_chol_ts[i] = __builtin_readcyclecounter(); g_chol_ts[64 + (i)] = wall_clock64(); } } while (0)
#else
#define DH_TS(i) do { } while (0)
#endif
constexpr int SB = 16;            // sub-block
constexpr int LDB = NB + 2;       // LDS leading dimension (doubles): 16 rows x {k, k+1} hit 32 distinct bank pairs

using f64x4 = __attribute__((ext_vector_type(4))) double;

__device__ __forceinline__ double lane_bcast(double v, int lane) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// 1/sqrt(a) in fp64: hardware estimate + two coupled Goldschmidt steps (g -> sqrt(a), h -> 1/(2 sqrt(a))): the
// dependent chain is rsq + 5 fma-class ops instead of rsq + 8 for the textbook Newton form
__device__ __forceinline__ double fast_rsqrt(double a) {
  const double y = __builtin_amdgcn_rsq(a);
  double g = a * y, h = 0.5 * y;
  double r = __builtin_fma(-g, h, 0.5);
  g = __builtin_fma(g, r, g); h = __builtin_fma(h, r, h);
  r = __builtin_fma(-g, h, 0.5);
  h = __builtin_fma(h, r, h);
  return 2.0 * h;
}

// 64x64 block of the system matrix <-> LDS tile (leading dimension LDB), all 16-byte loads in flight at once
__device__ __forceinline__ void load_block64(double* __restrict__ dst, const double* __restrict__ src, int ld, int tid) {
  double2 v[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) { const int o = tid + 256 * q; v[q] = *reinterpret_cast<const double2*>(src + (long)(o >> 5) * ld + (o & 31) * 2); }
#pragma unroll
  for (int q = 0; q < 8; ++q) { const int o = tid + 256 * q; *reinterpret_cast<double2*>(dst + (o >> 5) * LDB + (o & 31) * 2) = v[q]; }
}

// Blocked Cholesky of the stacked panel P = [A_jj ; A_rj] (128 x 64, LDS, leading dimension LDB) by 256 threads:
// on return rows 0..63 hold L_jj (lower triangle, zeros above) and rows 64..127 hold L_rj = A_rj L_jj^-T, i.e. the
// triangular solve of the block row comes out of the same elimination and no inverse is needed here.
// Per 8-column step: EVERY thread factorises the 8x8 diagonal sub-block redundantly in its own registers (pure
// register code with compile-time indices: no cross-lane traffic, no barrier before the next phase), one thread
// per row below substitutes against it (factor taken from registers), and the trailing columns are updated on
// the fp64 MFMA.  (8 columns per step: the scalar work of the diagonal factor grows with the cube of the step, the
// MFMA part does not: 16-column steps took 28 us per 64 columns, see profiles/.)  Returns false if a pivot is not
// positive and finite.
__device__ bool chol_panel128_lds(double* __restrict__ P, int tid) {
  constexpr int PB = 8;                                  // elimination step (columns per step); MFMA tiles stay 16x16
  const int lane = tid & 63, wave = tid >> 6;
  bool ok = true;
#pragma unroll 1
  for (int k = 0; k < NB / PB; ++k) {
    const int o = k * PB;
    double L[PB][PB];                                  // lower triangle used
    double dinv[PB];
#pragma unroll
    for (int r = 0; r < PB; ++r)
#pragma unroll
      for (int c = 0; c <= r; ++c) L[r][c] = P[(o + r) * LDB + o + c];       // same address in every lane: broadcast
#pragma unroll
    for (int c = 0; c < PB; ++c) {
      double piv = L[c][c];
      if (!(piv > 0.0) || !(piv < 1e300)) { ok = false; piv = 1.0; }
      const double rs = fast_rsqrt(piv);
      dinv[c] = rs;
      L[c][c] = piv * rs;
#pragma unroll
      for (int r = c + 1; r < PB; ++r) L[r][c] *= rs;
#pragma unroll
      for (int c2 = c + 1; c2 < PB; ++c2)
#pragma unroll
        for (int r = c2; r < PB; ++r) L[r][c2] -= L[r][c] * L[c2][c];
    }
    __syncthreads();                                   // everybody has read the sub-block
    if (tid < PB) {                                    // publish L_kk (thread r writes row r)
#pragma unroll
      for (int r = 0; r < PB; ++r)
        if (tid == r) {
#pragma unroll
          for (int c = 0; c < PB; ++c) P[(o + r) * LDB + o + c] = (c <= r) ? L[r][c] : 0.0;
        }
    }
    // rows below the sub-block (rest of A_jj and all of A_rj): X[i][:] = A[i][:] L_kk^-T, one thread per row
    const int nbelow = 2 * NB - o - PB;
    if (tid < nbelow) {
      double* prow = P + (o + PB + tid) * LDB + o;
      double x[PB];
#pragma unroll
      for (int c = 0; c < PB; ++c) x[c] = prow[c];
#pragma unroll
      for (int c = 0; c < PB; ++c) {
        double v = x[c];
#pragma unroll
        for (int m = 0; m < c; ++m) v -= x[m] * L[c][m];
        x[c] = v * dinv[c];
      }
#pragma unroll
      for (int c = 0; c < PB; ++c) prow[c] = x[c];
    }
    __syncthreads();
    // trailing update on the fp64 MFMA: P[i][j] -= sum_m X[i][m] X[j][m] over 16x16 tiles (tj = column tile of the
    // 64 columns, ti >= tj row tile of the 128 rows) that reach past row/column o + 8.  A tile that starts before
    // o + 8 (even k) also holds rows/columns <= the current step: their operands are taken as 0, so finished
    // entries are left untouched.
    const int first = o + PB;                          // first row/column still to be updated
    const int tj0 = first / SB, nCt = NB / SB - tj0, nRt = 2 * NB / SB - tj0;      // tiles from tj0 on
    const int ntiles = nCt * nRt - nCt * (nCt - 1) / 2;
    for (int t = wave; t < ntiles; t += 4) {
      int tj = 0, rem = t;
      while (rem >= nRt - tj) { rem -= nRt - tj; ++tj; }
      const int r0 = (tj0 + tj + rem) * SB, c0 = (tj0 + tj) * SB;
      f64x4 acc;
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = P[(r0 + (lane >> 4) + 4 * q) * LDB + c0 + (lane & 15)];
      const int ra = r0 + (lane & 15), rb = c0 + (lane & 15);
#pragma unroll
      for (int kk = 0; kk < PB / 4; ++kk) {
        const double av = ra >= first ? -P[ra * LDB + o + kk * 4 + (lane >> 4)] : 0.0;
        const double bv = rb >= first ? P[rb * LDB + o + kk * 4 + (lane >> 4)] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) P[(r0 + (lane >> 4) + 4 * q) * LDB + c0 + (lane & 15)] = acc[q];
    }
    __syncthreads();
  }
  return ok;
}

// 1/sqrt(a) for the pivot chain of the register panel: hardware estimate + ONE third-order step
// y (1 + e/2 + 3 e^2 / 8), e = 1 - a y^2 -- 5 dependent operations after the estimate instead of 7
__device__ __forceinline__ double pivot_rsqrt(double a) {
  const double y = __builtin_amdgcn_rsq(a);
  const double t = a * y;
  const double e = __builtin_fma(-t, y, 1.0);
  const double p = __builtin_fma(e, 0.375, 0.5);
  return __builtin_fma(y, p * e, y);
}

// The same stacked-panel factorisation with the UNFINISHED part of the panel held in registers as fp64-MFMA accumulator
// tiles: wave w owns rows 16w..16w+15 of A_jj (`top`) and of A_rj (`bot`), all four 16-column tiles (the layout
// mfma_abt_64 leaves the updated blocks in, so the look-ahead step never stages them).  P only carries what has to cross
// lanes: per 8-column step the owners drop the 8 current columns into P, every thread factors the 8x8 diagonal
// sub-block in its own registers, one thread per row (the 8 rows of the sub-block included: their substitution IS the
// row of L_kk) substitutes and writes the row of L back, and the trailing tiles take their rank-8 update straight from
// P into the accumulators -- first the tile that holds the NEXT 8 columns, whose columns go to P at once; the other
// tiles are updated after the barrier, interleaved with the next pivot chain (the fp64 MFMA of gfx950 takes 64 cycles;
// the chain leaves the issue slots for it).  Straight-line code: every wave issues the same MFMAs (a `top` tile above
// the diagonal gets a zero operand or is never read), so the accumulators never move between register files.
// Two barriers per step, no accumulator round trips through LDS.  On return P = [L_jj (zeros above the diagonal); L_rj].
__device__ __forceinline__ bool chol_panel128_regs(double* __restrict__ P, f64x4 (&top)[4], f64x4 (&bot)[4], int tid) {
  constexpr int PB = 8;
  const int lane = tid & 63, wave = tid >> 6, lr = lane >> 4, lc = lane & 15;
  bool ok = true;
  double aT[2] = {0.0, 0.0}, aB[2] = {0.0, 0.0}, bq[4][2] = {};
  if ((lc >> 3) == 0) {                                  // columns 0..7 of the panel
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = 16 * wave + lr + 4 * q;
      P[row * LDB + lc] = top[0][q];
      P[(NB + row) * LDB + lc] = bot[0][q];
    }
  }
#pragma unroll
  for (int k = 0; k < NB / PB; ++k) {
    const int o = k * PB, first = o + PB;
    const int nsub = 2 * NB - o;                         // rows o .. 127 are substituted (thread t <-> row o + t)
    __syncthreads();                                     // the step's 8 columns are in P
    DH_TS(8 + 4 * k);
    double L[PB][PB], dinv[PB], x[PB];
    // the sub-block as 16-byte broadcast reads at compile-time offsets (rows start 16-byte aligned: 528-byte row stride,
    // o a multiple of 8): 20 ds_read_b128 with immediate offsets instead of 36 separately addressed 8-byte reads
#pragma unroll
    for (int r = 0; r < PB; ++r)
#pragma unroll
      for (int c = 0; c <= r; c += 2) {
        const double2 v = *reinterpret_cast<const double2*>(P + (o + r) * LDB + o + c);
        L[r][c] = v.x;
        if (c + 1 <= r) L[r][c + 1] = v.y;
      }
    double* prow = P + (o + (tid < nsub ? tid : 0)) * LDB + o;
#pragma unroll
    for (int c = 0; c < PB; c += 2) {
      const double2 v = *reinterpret_cast<const double2*>(prow + c);
      x[c] = v.x; x[c + 1] = v.y;
    }
    // rest of the previous step's rank-8 update (column tiles after the one that holds this step's columns)
    if (k > 0) {
#pragma unroll
      for (int nt = (k >> 1) + 1; nt < 4; ++nt)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          bot[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(aB[kk], bq[nt][kk], bot[nt], 0, 0, 0);
          top[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(aT[kk], bq[nt][kk], top[nt], 0, 0, 0);
        }
    }
#pragma unroll
    for (int c = 0; c < PB; ++c) {
      const double piv = L[c][c];
      ok = ok & (piv > 0.0) & (piv < 1e300);               // off the dependent chain: a bad pivot only has to be reported
      const double rs = pivot_rsqrt(piv);
      dinv[c] = rs;
#pragma unroll
      for (int r = c + 1; r < PB; ++r) L[r][c] *= rs;
#pragma unroll
      for (int c2 = c + 1; c2 < PB; ++c2)
#pragma unroll
        for (int r = c2; r < PB; ++r) L[r][c2] = __builtin_fma(-L[r][c], L[c2][c], L[r][c2]);
    }
    // x <- x L_kk^-T (right-looking: one fma + one mul per column on the chain); for a row of the sub-block itself the
    // result is its row of L_kk, with zeros forced above the diagonal
#pragma unroll
    for (int c = 0; c < PB; ++c) {
      x[c] *= dinv[c];
#pragma unroll
      for (int m = c + 1; m < PB; ++m) x[m] = __builtin_fma(-x[c], L[m][c], x[m]);
    }
#pragma unroll
    for (int c = 1; c < PB; ++c) x[c] = tid < c ? 0.0 : x[c];
    if (k > 0) {                                         // one MFMA per ~14 chain operations
#pragma unroll
      for (int i = 0; i < 4 * (3 - (k >> 1)); ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 14, 0);
      }
    }
    DH_TS(9 + 4 * k);
    if (tid >= PB && tid < nsub) {
#pragma unroll
      for (int c = 0; c < PB; ++c) prow[c] = x[c];
    }
    __syncthreads();                                     // rows of L for this step are in P
    DH_TS(10 + 4 * k);
    if (tid < PB) {                                      // L_kk after the barrier: the others were still reading the sub-block
#pragma unroll
      for (int c = 0; c < PB; ++c) prow[c] = x[c];
    }
    if (k + 1 < NB / PB) {
      // operands of the rank-8 update: rows/columns before `first` are finished -> 0
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int rt = 16 * wave + lc;
        const double vt = P[rt * LDB + o + kk * 4 + lr];
        aT[kk] = rt >= first ? -vt : 0.0;
        aB[kk] = -P[(NB + rt) * LDB + o + kk * 4 + lr];
#pragma unroll
        for (int nt = first / SB; nt < 4; ++nt) {
          const int rb = 16 * nt + lc;
          const double vb = P[rb * LDB + o + kk * 4 + lr];
          bq[nt][kk] = rb >= first ? vb : 0.0;
        }
      }
      // the tile with the next 8 columns first; its columns go to P for the next step
      const int nu = first / SB, half = (first / PB) & 1;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        bot[nu] = __builtin_amdgcn_mfma_f64_16x16x4f64(aB[kk], bq[nu][kk], bot[nu], 0, 0, 0);
        top[nu] = __builtin_amdgcn_mfma_f64_16x16x4f64(aT[kk], bq[nu][kk], top[nu], 0, 0, 0);
      }
      if ((lc >> 3) == half) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = 16 * wave + lr + 4 * q;
          P[row * LDB + first + (lc & 7)] = row >= first ? top[nu][q] : 0.0;
          P[(NB + row) * LDB + first + (lc & 7)] = bot[nu][q];
        }
      }
    }
    DH_TS(11 + 4 * k);
  }
  __syncthreads();
  return ok;
}

// W = L^-1 (lower triangular, zero above the diagonal) from the factor in S; T = 3 x 16x16 scratch blocks
__device__ void tri_inverse64_lds(const double* __restrict__ S, const double* __restrict__ dinv, double* __restrict__ Wm,
                                  double* __restrict__ T, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  for (int idx = tid; idx < NB * NB; idx += 256) Wm[(idx >> 6) * LDB + (idx & 63)] = 0.0;
  __syncthreads();
  {
    // diagonal 16x16 blocks: wave k inverts block k, lane c = column c
    const int o = wave * SB, c = lane;
    if (c < SB) {
      double x[SB];
#pragma unroll
      for (int a = 0; a < SB; ++a) {
        double v = (a == c) ? 1.0 : 0.0;
#pragma unroll
        for (int m = 0; m < a; ++m) v -= S[(o + a) * LDB + o + m] * x[m];
        x[a] = (a < c) ? 0.0 : v * dinv[o + a];
      }
#pragma unroll
      for (int a = 0; a < SB; ++a) Wm[(o + a) * LDB + o + c] = x[a];
    }
  }
  __syncthreads();
  const int ti = tid >> 4, tj = tid & 15;                // one output element of a 16x16 block per thread
#pragma unroll 1
  for (int d = 1; d < NB / SB; ++d) {
    const int nblk = NB / SB - d;                        // blocks (k + d, k), k = 0 .. nblk-1
    for (int k = 0; k < nblk; ++k) {                     // T_k = sum_{m=k}^{k+d-1} L[k+d][m] W[m][k]
      const int i = k + d;
      double acc = 0.0;
      for (int m = k; m < i; ++m)
#pragma unroll
        for (int q = 0; q < SB; ++q) acc += S[(i * SB + ti) * LDB + m * SB + q] * Wm[(m * SB + q) * LDB + k * SB + tj];
      T[(k * SB + ti) * (SB + 1) + tj] = acc;
    }
    __syncthreads();
    for (int k = 0; k < nblk; ++k) {                     // W[k+d][k] = -W[k+d][k+d] T_k
      const int i = k + d;
      double acc = 0.0;
#pragma unroll
      for (int q = 0; q < SB; ++q) acc += Wm[(i * SB + ti) * LDB + i * SB + q] * T[(k * SB + q) * (SB + 1) + tj];
      Wm[(i * SB + ti) * LDB + k * SB + tj] = -acc;
    }
    __syncthreads();
  }
}

// D (64x64) = A B^T with A, B in LDS (leading dimension LDB); wave w owns rows 16w..16w+15.
// kmax4[nt] = number of 4-wide k steps for column tile nt (lets a triangular B skip its zero part).
// v_mfma_f64_16x16x4_f64: A operand lane l -> A[i = l&15][k = l>>4], B operand -> B^T[k][j] = B[j = l&15][k = l>>4],
// result reg q of lane l -> D[row = (l>>4) + 4q][col = l&15].
template <bool TRI>
__device__ __forceinline__ void mfma_abt_64(const double* __restrict__ A, const double* __restrict__ B, int wave, int lane,
                                            f64x4 (&acc)[4]) {
  const int i = wave * 16 + (lane & 15), kq = lane >> 4;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) acc[nt] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kk = 0; kk < NB / 4; ++kk) {
    const double a = A[i * LDB + kk * 4 + kq];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      if (TRI && kk >= 4 * (nt + 1)) continue;           // B[n][k] = 0 for k > n
      const double b = B[(nt * 16 + (lane & 15)) * LDB + kk * 4 + kq];
      acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[nt], 0, 0, 0);
    }
  }
}

// one 16x16 tile of A B^T: rows 16 rb .. of A against rows 16 nt .. of B (both 64 columns, LDS, leading dimension LDB)
__device__ __forceinline__ f64x4 mfma_abt_tile(const double* __restrict__ A, const double* __restrict__ B, int rb, int nt, int lane) {
  f64x4 acc = {0.0, 0.0, 0.0, 0.0};
  const int kq = lane >> 4;
#pragma unroll
  for (int kk = 0; kk < NB / 4; ++kk)
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(rb * 16 + (lane & 15)) * LDB + kk * 4 + kq], B[(nt * 16 + (lane & 15)) * LDB + kk * 4 + kq],
                                               acc, 0, 0, 0);
  return acc;
}

__global__ __launch_bounds__(256, 1) void chol_panel_kernel(double* __restrict__ H, int ld, int j, int* __restrict__ meta,
                                                            double* __restrict__ Ldiag) {
  extern __shared__ double s_chol[];
  double* P = s_chol;                     // [128][LDB]: A_jj over A_rj
  const int tid = threadIdx.x;
  const int r = j + 1 + blockIdx.x;
  const long d0 = (long)j * NB, r0 = (long)r * NB;
  {                                          // both blocks in flight together
    double2 va[8], vb[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int o = tid + 256 * q;
      va[q] = *reinterpret_cast<const double2*>(H + (d0 + (o >> 5)) * ld + d0 + (o & 31) * 2);
      vb[q] = *reinterpret_cast<const double2*>(H + (r0 + (o >> 5)) * ld + d0 + (o & 31) * 2);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int o = tid + 256 * q;
      *reinterpret_cast<double2*>(P + (o >> 5) * LDB + (o & 31) * 2) = va[q];
      *reinterpret_cast<double2*>(P + (NB + (o >> 5)) * LDB + (o & 31) * 2) = vb[q];
    }
  }
  __syncthreads();
  const bool ok = chol_panel128_lds(P, tid);
  if (!ok && tid == 0) meta[1] = 1;
  // L_rj back to the matrix.  Workgroup 0 also publishes L_jj -- into Ldiag, NOT in place: the other workgroups of
  // this launch read A_jj from H at their own pace (no grid-level ordering), and only chol_inverse_kernel needs L_jj
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int o = tid + 256 * q;
    *reinterpret_cast<double2*>(H + (r0 + (o >> 5)) * ld + d0 + (o & 31) * 2) = *reinterpret_cast<const double2*>(P + (NB + (o >> 5)) * LDB + (o & 31) * 2);
  }
  if (blockIdx.x == 0) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int o = tid + 256 * q;
      *reinterpret_cast<double2*>(Ldiag + (long)j * NB * NB + (o >> 5) * NB + (o & 31) * 2) = *reinterpret_cast<const double2*>(P + (o >> 5) * LDB + (o & 31) * 2);
    }
  }
}

// W_j = L_jj^-1 for every diagonal block (needed by the back substitution only): one workgroup per block, after
// the factorisation, off its critical path
__global__ __launch_bounds__(256) void chol_inverse_kernel(const double* __restrict__ Ldiag, double* __restrict__ Linv) {
  extern __shared__ double s_chol[];
  double* S = s_chol;
  double* Wm = S + NB * LDB;
  double* T = Wm + NB * LDB;
  double* dinv = T + 3 * SB * (SB + 1);
  const int tid = threadIdx.x, j = blockIdx.x;
  load_block64(S, Ldiag + (long)j * NB * NB, NB, tid);
  __syncthreads();
  if (tid < NB) dinv[tid] = 1.0 / S[tid * LDB + tid];
  __syncthreads();
  tri_inverse64_lds(S, dinv, Wm, T, tid);
  for (int o = tid; o < NB * NB; o += 256) Linv[(long)j * NB * NB + o] = Wm[(o >> 6) * LDB + (o & 63)];
}

// trailing update A_rc -= L_rj L_cj^T for j < c <= r (r runs over the rhs block row too)
__global__ __launch_bounds__(256) void chol_update_kernel(double* __restrict__ H, int ld, int j, int nbk) {
  const int r = j + 1 + blockIdx.y, c = j + 1 + blockIdx.x;
  if (c > r || c >= nbk) return;
  extern __shared__ double s_chol[];
  double* sA = s_chol;
  double* sB = sA + NB * LDB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long r0 = (long)r * NB, c0 = (long)c * NB, d0 = (long)j * NB;
  // both operand blocks and the block to be updated are requested in one burst
  double2 va[8], vb[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int o = tid + 256 * q;
    va[q] = *reinterpret_cast<const double2*>(H + (r0 + (o >> 5)) * ld + d0 + (o & 31) * 2);
    vb[q] = *reinterpret_cast<const double2*>(H + (c0 + (o >> 5)) * ld + d0 + (o & 31) * 2);
  }
  double cur[4][4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int q = 0; q < 4; ++q) cur[nt][q] = H[(r0 + wave * 16 + (lane >> 4) + 4 * q) * ld + c0 + nt * 16 + (lane & 15)];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int o = tid + 256 * q;
    *reinterpret_cast<double2*>(sA + (o >> 5) * LDB + (o & 31) * 2) = va[q];
    *reinterpret_cast<double2*>(sB + (o >> 5) * LDB + (o & 31) * 2) = vb[q];
  }
  __syncthreads();
  f64x4 acc[4];
  mfma_abt_64<false>(sA, sB, wave, lane, acc);
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      H[(r0 + wave * 16 + (lane >> 4) + 4 * q) * ld + c0 + nt * 16 + (lane & 15)] = cur[nt][q] - acc[nt][q];
}

// One launch per block column from the second on ("look-ahead"): step j applies column j to the rest of the matrix and,
// in the same launch, factors column j+1 -- the panel no longer waits for the whole trailing update of the previous
// column, only for its own two blocks.
//   workgroups 0 .. nP-1 (block rows r = j+2 .. incl. the rhs row):  A_{j+1,j+1} -= L_{j+1,j} L_{j+1,j}^T and
//       A_{r,j+1} -= L_{r,j} L_{j+1,j}^T on the fp64 MFMA, written straight into the stacked LDS panel, then the
//       same stacked factorisation as chol_panel_kernel -> L_{r,j+1} (workgroup 0 also stores L_{j+1,j+1});
//   the others: A_rc -= L_rj L_cj^T for j+2 <= c <= r (chol_update_kernel's work minus block column j+1).
template <bool REGP>
__global__ __launch_bounds__(256, 1) void chol_step_kernel(double* __restrict__ H, int ld, int j, int nbk, int nP,
                                                           int* __restrict__ meta, double* __restrict__ Ldiag) {
  extern __shared__ double s_chol[];
  double* sA = s_chol;                    // L_rj, later rows 64..127 of the stacked panel
  double* sB = sA + NB * LDB;             // L_cj
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool panel = (int)blockIdx.x < nP;
  DH_TS(0);
  int r, c;
  if (panel) { r = j + 2 + blockIdx.x; c = j + 1; }
  else {
    // block rows / columns j+2 .. j+1+nP, lower triangle only: u = rr (rr + 1) / 2 + cc, cc <= rr (a square grid would
    // start nP^2 workgroups of which half return at once -- each still takes a 72-KB LDS slot from the ones that work)
    const int u = blockIdx.x - nP;
    int rr = (int)((sqrtf(8.f * (float)u + 1.f) - 1.f) * 0.5f);
    while (rr * (rr + 1) / 2 > u) --rr;
    while ((rr + 1) * (rr + 2) / 2 <= u) ++rr;
    const int cc = u - rr * (rr + 1) / 2;
    r = j + 2 + rr; c = j + 2 + cc;
    if (c >= nbk) return;                              // (the rhs block row has no diagonal block)
  }
  const long r0 = (long)r * NB, c0 = (long)c * NB, d0 = (long)j * NB;
  double2 va[8], vb[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int o = tid + 256 * q;
    va[q] = *reinterpret_cast<const double2*>(H + (r0 + (o >> 5)) * ld + d0 + (o & 31) * 2);
    vb[q] = *reinterpret_cast<const double2*>(H + (c0 + (o >> 5)) * ld + d0 + (o & 31) * 2);
  }
  double cur[4][4], curd[4][4];             // A_rc and, for a panel workgroup, the diagonal block A_cc
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      cur[nt][q] = H[(r0 + wave * 16 + (lane >> 4) + 4 * q) * ld + c0 + nt * 16 + (lane & 15)];
      if (panel) curd[nt][q] = H[(c0 + wave * 16 + (lane >> 4) + 4 * q) * ld + c0 + nt * 16 + (lane & 15)];
    }
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int o = tid + 256 * q;
    *reinterpret_cast<double2*>(sA + (o >> 5) * LDB + (o & 31) * 2) = va[q];
    *reinterpret_cast<double2*>(sB + (o >> 5) * LDB + (o & 31) * 2) = vb[q];
  }
  __syncthreads();
  DH_TS(1);
  f64x4 acc[4];
  mfma_abt_64<false>(sA, sB, wave, lane, acc);
  if (!panel) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        H[(r0 + wave * 16 + (lane >> 4) + 4 * q) * ld + c0 + nt * 16 + (lane & 15)] = cur[nt][q] - acc[nt][q];
    return;
  }
  // A_cc -= L_cj L_cj^T: only the tiles on and below the diagonal are ever used (wave w: column tiles 0..w) and the fp64 MFMA
  // (64 cycles) is the whole cost of this phase, so the 10 tiles are spread 2 / 2 / 3 / 3 over the waves: wave 0 also computes
  // wave 3's tile 0 and hands it over through LDS -- 112 instead of 128 MFMAs on the longest wave
  f64x4 accd[4];
  double* const xch = s_chol + 2 * NB * LDB;       // 256 doubles behind the two operand blocks
  if (REGP) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) accd[nt] = f64x4{0.0, 0.0, 0.0, 0.0};
    if (wave == 0) {
      accd[0] = mfma_abt_tile(sB, sB, 0, 0, lane);
      const f64x4 t30 = mfma_abt_tile(sB, sB, 3, 0, lane);
#pragma unroll
      for (int q = 0; q < 4; ++q) xch[q * 64 + lane] = t30[q];
    } else if (wave == 1) {
      accd[0] = mfma_abt_tile(sB, sB, 1, 0, lane); accd[1] = mfma_abt_tile(sB, sB, 1, 1, lane);
    } else if (wave == 2) {
      accd[0] = mfma_abt_tile(sB, sB, 2, 0, lane); accd[1] = mfma_abt_tile(sB, sB, 2, 1, lane); accd[2] = mfma_abt_tile(sB, sB, 2, 2, lane);
    } else {
      accd[1] = mfma_abt_tile(sB, sB, 3, 1, lane); accd[2] = mfma_abt_tile(sB, sB, 3, 2, lane); accd[3] = mfma_abt_tile(sB, sB, 3, 3, lane);
    }
  } else {
    mfma_abt_64<false>(sB, sB, wave, lane, accd);
  }
  __syncthreads();                          // every wave has read both operand blocks: the panel may overwrite them
  if (REGP && wave == 3) {
#pragma unroll
    for (int q = 0; q < 4; ++q) accd[0][q] = xch[q * 64 + lane];
  }
  DH_TS(2);
  double* P = s_chol;                       // [128][LDB]: updated A_cc over updated A_rc
  bool ok;
  if (REGP) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int q = 0; q < 4; ++q) { accd[nt][q] = curd[nt][q] - accd[nt][q]; acc[nt][q] = cur[nt][q] - acc[nt][q]; }
    ok = chol_panel128_regs(P, accd, acc, tid);
  } else {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = wave * 16 + (lane >> 4) + 4 * q, col = nt * 16 + (lane & 15);
        P[row * LDB + col] = curd[nt][q] - accd[nt][q];
        P[(NB + row) * LDB + col] = cur[nt][q] - acc[nt][q];
      }
    __syncthreads();
    ok = chol_panel128_lds(P, tid);
  }
  if (!ok && tid == 0) meta[1] = 1;
  DH_TS(3);
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int o = tid + 256 * q;
    *reinterpret_cast<double2*>(H + (r0 + (o >> 5)) * ld + c0 + (o & 31) * 2) = *reinterpret_cast<const double2*>(P + (NB + (o >> 5)) * LDB + (o & 31) * 2);
  }
  if (blockIdx.x == 0) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int o = tid + 256 * q;
      *reinterpret_cast<double2*>(Ldiag + (long)c * NB * NB + (o >> 5) * NB + (o & 31) * 2) = *reinterpret_cast<const double2*>(P + (o >> 5) * LDB + (o & 31) * 2);
    }
  }
  DH_TS(4);
}

#ifdef DH_ABLATION
// ---- dataflow schedule (chol_lookahead = 2, -DDH_ABLATION builds): ONE persistent launch for the whole factorisation -------------
// The look-ahead schedule pays a kernel boundary per block column: 47 dependent launches of 18-33 us at 512 keyframes, of which the
// panel itself is a third.  Here the launches' implicit grid barriers are replaced by one ready flag per block of L.  A task = one
// block (r, c) below the diagonal (the rhs block row included), LEFT-looking: it takes A_rc and A_cc, applies
// A_rc -= L_rj L_cj^T and A_cc -= L_cj L_cj^T for j = 0 .. c-1 AS THE FLAGS OF (r, j) AND (c, j) COME UP (same updates in the same
// order as the launches apply them: identical factors), then runs the stacked panel factorisation of the look-ahead step and raises
// its own flag.  Workgroups draw tasks from a ticket counter in column-major order, so everything a task waits for belongs to a ticket
// drawn earlier by a workgroup that is running: no deadlock whatever the number of resident workgroups.  flow[0] = ticket counter,
// flow[1] = watchdog word (a wait that does not end raises it, which ends every wait and fails the factorisation like a bad pivot
// instead of hanging the device), flow[FLOW_HDR + r * nbk + c] = 1 when L_rc is in H.  Polls and flag stores are read-modify-writes
// (executed at the memory side: never served from one XCD's own L2); `zero` = 0, a kernel argument the compiler cannot fold.
// All LDS is dynamic (the task word sits behind the panel): the panel's 16-byte accesses need the dynamic base at offset 0 -- a 4-byte
// static __shared__ word in front of it made the first version of this kernel hang the queue.
// Measured at 512 keyframes (profiles/r04_w_chol_dataflow.txt): results identical to the look-ahead schedule bit for bit, 3.67 vs 3.51 ms
// per global BA -- a release / acquire pair across the eight L2s per block column costs what a kernel boundary costs.  Not faster
// yet: the acquires could be batched (one per group of finished columns) and the next update's blocks prefetched.
__global__ __launch_bounds__(256, 1) void chol_flow_kernel(double* __restrict__ H, int ld, int nbk, int* __restrict__ meta,
                                                           double* __restrict__ Ldiag, int* __restrict__ flow, int zero) {
  extern __shared__ double s_chol[];
  double* sA = s_chol;                    // L_rj, later rows 64..127 of the stacked panel
  double* sB = sA + NB * LDB;             // L_cj
  double* const xch = s_chol + 2 * NB * LDB;
  volatile int* s_task = reinterpret_cast<volatile int*>(xch + 256);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntasks = nbk * (nbk + 1) / 2;
  int* const flags = flow + FLOW_HDR;
  for (int round = 0; round <= ntasks; ++round) {
    if (tid == 0) s_task[0] = atomicAdd(&flow[0], 1);
    __syncthreads();
    const int t = s_task[0];
    __syncthreads();                        // (the task word is rewritten by the next round)
    if (t >= ntasks) return;
    int c = 0, rem = t;
    while (c < nbk && rem >= nbk - c) { rem -= nbk - c; ++c; }       // column c has the block rows c+1 .. nbk (nbk = the rhs row)
    const int r = c + 1 + rem;
    const long r0 = (long)r * NB, c0 = (long)c * NB;
    double cur[4][4], curd[4][4];           // A_rc and A_cc in the accumulator layout of mfma_abt_64
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        cur[nt][q] = H[(r0 + wave * 16 + (lane >> 4) + 4 * q) * ld + c0 + nt * 16 + (lane & 15)];
        curd[nt][q] = H[(c0 + wave * 16 + (lane >> 4) + 4 * q) * ld + c0 + nt * 16 + (lane & 15)];
      }
    for (int j = 0; j < c; ++j) {
      if (tid == 0) {
        int spins = 0;
        while (atomicAdd(&flags[r * nbk + j], zero) == 0 || atomicAdd(&flags[c * nbk + j], zero) == 0) {
          __builtin_amdgcn_s_sleep(2);
          if ((++spins & 63) == 0 && (spins > (1 << 16) || atomicAdd(&flow[1], zero) != 0)) { atomicExch(&flow[1], 1); meta[1] = 1; break; }
        }
      }
      __syncthreads();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // the two blocks were written by other workgroups (other XCDs, other L2s)
      const long d0 = (long)j * NB;
      double2 va[8], vb[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int o = tid + 256 * q;
        va[q] = *reinterpret_cast<const double2*>(H + (r0 + (o >> 5)) * ld + d0 + (o & 31) * 2);
        vb[q] = *reinterpret_cast<const double2*>(H + (c0 + (o >> 5)) * ld + d0 + (o & 31) * 2);
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int o = tid + 256 * q;
        *reinterpret_cast<double2*>(sA + (o >> 5) * LDB + (o & 31) * 2) = va[q];
        *reinterpret_cast<double2*>(sB + (o >> 5) * LDB + (o & 31) * 2) = vb[q];
      }
      __syncthreads();
      f64x4 acc[4], accd[4];
      mfma_abt_64<false>(sA, sB, wave, lane, acc);
      // A_cc -= L_cj L_cj^T, lower tiles only, spread 2 / 2 / 3 / 3 over the waves (see chol_step_kernel)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) accd[nt] = f64x4{0.0, 0.0, 0.0, 0.0};
      if (wave == 0) {
        accd[0] = mfma_abt_tile(sB, sB, 0, 0, lane);
        const f64x4 t30 = mfma_abt_tile(sB, sB, 3, 0, lane);
#pragma unroll
        for (int q = 0; q < 4; ++q) xch[q * 64 + lane] = t30[q];
      } else if (wave == 1) {
        accd[0] = mfma_abt_tile(sB, sB, 1, 0, lane); accd[1] = mfma_abt_tile(sB, sB, 1, 1, lane);
      } else if (wave == 2) {
        accd[0] = mfma_abt_tile(sB, sB, 2, 0, lane); accd[1] = mfma_abt_tile(sB, sB, 2, 1, lane); accd[2] = mfma_abt_tile(sB, sB, 2, 2, lane);
      } else {
        accd[1] = mfma_abt_tile(sB, sB, 3, 1, lane); accd[2] = mfma_abt_tile(sB, sB, 3, 2, lane); accd[3] = mfma_abt_tile(sB, sB, 3, 3, lane);
      }
      __syncthreads();                      // every wave has read both operand blocks (next j / the panel may overwrite them)
      if (wave == 3) {
#pragma unroll
        for (int q = 0; q < 4; ++q) accd[0][q] = xch[q * 64 + lane];
      }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) { curd[nt][q] -= accd[nt][q]; cur[nt][q] -= acc[nt][q]; }
    }
    f64x4 top[4], bot[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int q = 0; q < 4; ++q) { top[nt][q] = curd[nt][q]; bot[nt][q] = cur[nt][q]; }
    double* P = s_chol;                     // [128][LDB]: A_cc over A_rc, both fully updated
    const bool ok = chol_panel128_regs(P, top, bot, tid);
    if (!ok && tid == 0) meta[1] = 1;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int o = tid + 256 * q;
      *reinterpret_cast<double2*>(H + (r0 + (o >> 5)) * ld + c0 + (o & 31) * 2) = *reinterpret_cast<const double2*>(P + (NB + (o >> 5)) * LDB + (o & 31) * 2);
    }
    if (r == c + 1) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int o = tid + 256 * q;
        *reinterpret_cast<double2*>(Ldiag + (long)c * NB * NB + (o >> 5) * NB + (o & 31) * 2) = *reinterpret_cast<const double2*>(P + (o >> 5) * LDB + (o & 31) * 2);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");    // the block is on its way to memory before the flag
    __syncthreads();                                      // (also: the panel is read, the next task may overwrite it)
    if (tid == 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); atomicExch(&flags[r * nbk + c], 1); }
  }
}

// ---- dataflow schedule, second form (chol_lookahead = 3; measured: 3.70 ms per global BA against 3.66 for the first form, 3.47 look-ahead)
// What the measurement of the first form asks for: an update costs ~9 us there (acquire fence, two cold 32-KB block
// loads through registers, 110 MFMAs, three barriers) whether or not its operands have been ready for a long time.  Here
//   * ONE acquire and ONE vector read of the flags (wave 0, a lane per column) per GROUP of columns whose blocks are already flagged;
//   * the operand blocks arrive by LDS-DMA (global_load_lds_dwordx4, one 512-byte matrix row per half-wave instruction into the
//     padded LDS rows): no staging registers, and the next update's blocks land in the second LDS buffer under this update's MFMAs.
// Same updates in the same order as the other schedules: results identical bit for bit (scripts/bench_chol.py; profiles/r04_zz_chol_schedules_ab.txt).
// It is not faster: the time is the chain through flag, acquire, cold loads, MFMAs, panel, store and release, not the early updates.
__global__ __launch_bounds__(256, 1) void chol_flow_dma_kernel(double* __restrict__ H, int ld, int nbk, int* __restrict__ meta,
                                                               double* __restrict__ Ldiag, int* __restrict__ flow, int zero) {
  extern __shared__ double s_chol[];
  constexpr int BUF = 2 * NB * LDB;                                  // doubles per operand buffer: L_rj over L_cj
  double* const xch = s_chol + 2 * BUF;
  volatile int* s_task = reinterpret_cast<volatile int*>(xch + 256);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntasks = nbk * (nbk + 1) / 2;
  int* const flags = flow + FLOW_HDR;
  const unsigned lds0 = (unsigned)(uintptr_t)s_chol;
  for (int round = 0; round <= ntasks; ++round) {
    if (tid == 0) s_task[0] = atomicAdd(&flow[0], 1);
    __syncthreads();
    const int t = s_task[0];
    __syncthreads();
    if (t >= ntasks) return;
    int c = 0, rem = t;
    while (c < nbk && rem >= nbk - c) { rem -= nbk - c; ++c; }
    const int r = c + 1 + rem;
    const long r0 = (long)r * NB, c0 = (long)c * NB;
    double cur[4][4], curd[4][4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        cur[nt][q] = H[(r0 + wave * 16 + (lane >> 4) + 4 * q) * ld + c0 + nt * 16 + (lane & 15)];
        curd[nt][q] = H[(c0 + wave * 16 + (lane >> 4) + 4 * q) * ld + c0 + nt * 16 + (lane & 15)];
      }
    // blocks (r, j) and (c, j) -> buffer `b`: wave w moves rows 16 w .. 16 w + 15 of both, one row (512 B = 32 lanes x 16 B) per instruction
    auto dma = [&](int j, int b) {
      const long d0 = (long)j * NB;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const long blk0 = half ? c0 : r0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = wave * 16 + i;
          const unsigned long gaddr = (unsigned long)(H + (blk0 + row) * ld + d0);
          const unsigned glo = __builtin_amdgcn_readfirstlane((unsigned)gaddr);
          const unsigned ghi = __builtin_amdgcn_readfirstlane((unsigned)(gaddr >> 32));
          const void* gs = reinterpret_cast<const void*>(((unsigned long)ghi << 32) | glo);
          const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(((long)b * BUF + (long)half * NB * LDB + (long)row * LDB) * sizeof(double)));
          if (lane < 32) {
            unsigned keep_;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep_) : "v"(lane * 16), "s"(gs), "s"(dst) : "memory");
          }
        }
      }
    };
    auto wait_for = [&](int j) {
      int spins = 0;
      while (atomicAdd(&flags[r * nbk + j], zero) == 0 || atomicAdd(&flags[c * nbk + j], zero) == 0) {
        __builtin_amdgcn_s_sleep(2);
        if ((++spins & 63) == 0 && (spins > (1 << 16) || atomicAdd(&flow[1], zero) != 0)) { atomicExch(&flow[1], 1); meta[1] = 1; break; }
      }
    };
    int b = 0;
    for (int j = 0; j < c;) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      if (wave == 0) {
        const int jj = j + lane;
        const bool up = jj < c && __hip_atomic_load(&flags[r * nbk + jj], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 &&
                        __hip_atomic_load(&flags[c * nbk + jj], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        const unsigned long long m = __ballot(up);
        int n = m == ~0ull ? 64 : __builtin_ctzll(~m);                    // leading columns whose blocks are flagged
        if (lane == 0) {
          if (n == 0) { wait_for(j); n = 1; }
          s_task[1] = n;
        }
      }
      __syncthreads();
      const int k = s_task[1];
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                  // (after a wait: the blocks flagged meanwhile)
      dma(j, b);
      for (int u = 0; u < k; ++u) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // this wave's rows of update u are in LDS ...
        __syncthreads();                                                  // ... and every other wave's
        if (u + 1 < k) dma(j + u + 1, b ^ 1);                             // the next update's blocks land under the MFMAs
        const double* sA = s_chol + (long)b * BUF;
        const double* sB = sA + NB * LDB;
        f64x4 acc[4], accd[4];
        mfma_abt_64<false>(sA, sB, wave, lane, acc);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) accd[nt] = f64x4{0.0, 0.0, 0.0, 0.0};
        if (wave == 0) {
          accd[0] = mfma_abt_tile(sB, sB, 0, 0, lane);
          const f64x4 t30 = mfma_abt_tile(sB, sB, 3, 0, lane);
#pragma unroll
          for (int q = 0; q < 4; ++q) xch[q * 64 + lane] = t30[q];
        } else if (wave == 1) {
          accd[0] = mfma_abt_tile(sB, sB, 1, 0, lane); accd[1] = mfma_abt_tile(sB, sB, 1, 1, lane);
        } else if (wave == 2) {
          accd[0] = mfma_abt_tile(sB, sB, 2, 0, lane); accd[1] = mfma_abt_tile(sB, sB, 2, 1, lane); accd[2] = mfma_abt_tile(sB, sB, 2, 2, lane);
        } else {
          accd[1] = mfma_abt_tile(sB, sB, 3, 1, lane); accd[2] = mfma_abt_tile(sB, sB, 3, 2, lane); accd[3] = mfma_abt_tile(sB, sB, 3, 3, lane);
        }
        __syncthreads();                      // every wave has read buffer b (and xch is written)
        if (wave == 3) {
#pragma unroll
          for (int q = 0; q < 4; ++q) accd[0][q] = xch[q * 64 + lane];
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int q = 0; q < 4; ++q) { curd[nt][q] -= accd[nt][q]; cur[nt][q] -= acc[nt][q]; }
        b ^= 1;
      }
      j += k;
    }
    f64x4 top[4], bot[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int q = 0; q < 4; ++q) { top[nt][q] = curd[nt][q]; bot[nt][q] = cur[nt][q]; }
    double* P = s_chol;                       // [128][LDB] = operand buffer 0 (every wave is past its last read of both buffers)
    const bool ok = chol_panel128_regs(P, top, bot, tid);
    if (!ok && tid == 0) meta[1] = 1;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int o = tid + 256 * q;
      *reinterpret_cast<double2*>(H + (r0 + (o >> 5)) * ld + c0 + (o & 31) * 2) = *reinterpret_cast<const double2*>(P + (NB + (o >> 5)) * LDB + (o & 31) * 2);
    }
    if (r == c + 1) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int o = tid + 256 * q;
        *reinterpret_cast<double2*>(Ldiag + (long)c * NB * NB + (o >> 5) * NB + (o & 31) * 2) = *reinterpret_cast<const double2*>(P + (o >> 5) * LDB + (o & 31) * 2);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (tid == 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); atomicExch(&flags[r * nbk + c], 1); }
  }
}

#endif  // DH_ABLATION

// back substitution of L^T x = y (y = row `brow` of H) in groups of BG block rows per launch:
// every workgroup solves the group's triangular system itself (x_g = W_g^T (y_g - sum_{g' > g in group} L_g'g^T x_g')),
// workgroup 0 publishes x, workgroup 1 + q applies the group's contribution to y_q for its own q < group start.
// Everything a workgroup needs (the W blocks, the in-group L block, its own L blocks for the final update) is
// requested in one burst up front, so the dependent phases run out of LDS / registers.
constexpr int BG = 2;
__global__ __launch_bounds__(256) void chol_backsub_kernel(double* __restrict__ H, int ld, int jhi, int brow,
                                                           const double* __restrict__ Linv, double* __restrict__ x) {
  extern __shared__ double s_chol[];
  double* sW = s_chol;                    // [BG][NB][LDB]
  double* sL = sW + BG * NB * LDB;        // block (jhi, jlo) of L when the group has two block rows
  __shared__ double sx[BG * NB];
  __shared__ double sy[BG * NB];
  __shared__ double sp[4][NB];
  const int tid = threadIdx.x;
  const int jlo = max(0, jhi - BG + 1), ng = jhi - jlo + 1;
  const int col = tid & 63, part = tid >> 6;
  for (int g = 0; g < ng; ++g) load_block64(sW + g * NB * LDB, Linv + (long)(jlo + g) * NB * NB, NB, tid);
  if (ng == 2) load_block64(sL, H + (long)jhi * NB * ld + (long)jlo * NB, ld, tid);
  const int q = (int)blockIdx.x - 1;                    // earlier block row updated by this workgroup
  const bool upd = q >= 0 && q < jlo;
  double lq[BG][16];
#pragma unroll
  for (int g = 0; g < BG; ++g)
#pragma unroll
    for (int a = 0; a < 16; ++a)
      lq[g][a] = (upd && g < ng) ? H[((long)(jlo + g) * NB + part * 16 + a) * ld + (long)q * NB + col] : 0.0;
  for (int o = tid; o < ng * NB; o += 256) sy[o] = H[(long)brow * ld + (long)jlo * NB + o];
  __syncthreads();
  for (int g = ng - 1; g >= 0; --g) {
    // x_g = W_g^T y_g : x[c] = sum_{m >= c} W[m][c] y[m]; 4 partial sums over m
    {
      double sacc = 0.0;
      const double* Wg = sW + g * NB * LDB;
#pragma unroll
      for (int m = part * 16; m < part * 16 + 16; ++m) sacc += Wg[m * LDB + col] * sy[g * NB + m];
      sp[part][col] = sacc;
    }
    __syncthreads();
    if (tid < NB) {
      const double v = (sp[0][tid] + sp[1][tid]) + (sp[2][tid] + sp[3][tid]);
      sx[g * NB + tid] = v;
      if (blockIdx.x == 0) x[(long)(jlo + g) * NB + tid] = v;
    }
    __syncthreads();
    if (g == 1) {                                       // y_0 -= L[jhi][jlo]^T x_1
      double sacc = 0.0;
#pragma unroll
      for (int a = part * 16; a < part * 16 + 16; ++a) sacc += sL[a * LDB + col] * sx[NB + a];
      sp[part][col] = sacc;
      __syncthreads();
      if (tid < NB) sy[tid] -= (sp[0][tid] + sp[1][tid]) + (sp[2][tid] + sp[3][tid]);
      __syncthreads();
    }
  }
  if (!upd) return;
  double sacc = 0.0;
#pragma unroll
  for (int g = 0; g < BG; ++g)
#pragma unroll
    for (int a = 0; a < 16; ++a) sacc += lq[g][a] * (g < ng ? sx[g * NB + part * 16 + a] : 0.0);
  sp[part][col] = sacc;
  __syncthreads();
  if (tid < NB) H[(long)brow * ld + (long)q * NB + tid] -= (sp[0][tid] + sp[1][tid]) + (sp[2][tid] + sp[3][tid]);
}

__global__ void ba_dx_kernel(const double* __restrict__ x, const int* __restrict__ meta, int n,
                             float* __restrict__ dx_ws, float* __restrict__ dx_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = (meta[1] | meta[2]) ? 0.f : (float)x[i];
  dx_ws[i] = v;
  if (dx_out) dx_out[i] = v;
}

// ------------------------------------------------------------------------------------------ update
__global__ __launch_bounds__(256) void ba_backsub_kernel(
    float* __restrict__ disps, const float* __restrict__ Q, const float* __restrict__ W,
    const float* __restrict__ Ei, const float* __restrict__ Ej, const float* __restrict__ dx,
    const int64_t* __restrict__ jj, const int* __restrict__ kx, const int* __restrict__ eoff,
    const int* __restrict__ eidx, const int* __restrict__ meta, int HW, int t0, int P,
    float* __restrict__ dz_out, int n_dz_rows, int own_lo, int own_hi) {
  const int k = blockIdx.x;
  const int p = blockIdx.y * 256 + threadIdx.x;
  if (p >= HW) return;
  if (k >= meta[0]) { if (dz_out && k < n_dz_rows) dz_out[(long)k * HW + p] = 0.f; return; }
  const int f = kx[k];
  float s = 0.f;
  {
    const int r = f - t0;
    if (r > 0 && r < P) {                                   // rows with relative pose index <= 0 are skipped
#pragma unroll
      for (int a = 0; a < 6; ++a) s += Ei[((long)k * 6 + a) * HW + p] * dx[6 * r + a];
    }
  }
  const int e0 = eoff[k], e1 = eoff[k + 1];
  for (int ei = e0; ei < e1; ++ei) {
    const int e = eidx[ei];
    const int r = (int)jj[e] - t0;
    if (r > 0 && r < P) {
#pragma unroll
      for (int a = 0; a < 6; ++a) s += Ej[((long)e * 6 + a) * HW + p] * dx[6 * r + a];
    }
  }
  // flagged arguments: no update at all; edge-sharded solver: only the owner of a frame moves its depths (a non-owner
  // holds none of the frame's edges, its Q / w are the damping and the sensor prior alone)
  const float dz = (meta[2] || f < own_lo || f >= own_hi) ? 0.f : Q[(long)k * HW + p] * (W[(long)k * HW + p] - s);
  disps[(long)f * HW + p] += dz;
  if (dz_out && k < n_dz_rows) dz_out[(long)k * HW + p] = dz;
}

__global__ void ba_pose_retr_kernel(float* __restrict__ poses, const float* __restrict__ dx, int t0, int P) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  float xi[6];
#pragma unroll
  for (int a = 0; a < 6; ++a) xi[a] = dx[6 * i + a];
  SE3f T = load_pose(poses + 7 * (long)(t0 + i));
  store_pose(poses + 7 * (long)(t0 + i), retr(xi, T));
}


// ------------------------------------------------------------------------------------------ edge-sharded exchange
// The reduced camera system of one rank's edges is non-zero only on the co-visible 6x6 blocks (lower triangle).  The
// multi-GPU solver exchanges ONE contiguous fp64 buffer per Gauss-Newton iteration:
//   [n_blocks][6][6] blocks (bp[b], bq[b]) | rhs [6P] | status[2]
// status[0] = this rank's argument flag (meta[2] != 0), status[1] = a flag of the host (block pattern stale); both are
// summed by the all-reduce, so every rank sees the same verdict and applies / skips the update together.
__global__ __launch_bounds__(256) void ba_pack_blocks_kernel(
    const double* __restrict__ H, int ld, int npad, int n, const int* __restrict__ bp, const int* __restrict__ bq,
    int n_blocks, const int* __restrict__ meta, int host_flags, double* __restrict__ packed) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long nb36 = (long)n_blocks * 36;
  if (i < nb36) {
    const int b = (int)(i / 36), rc = (int)(i - (long)b * 36), r = rc / 6, c = rc - 6 * r;
    packed[i] = H[(long)(6 * bp[b] + r) * ld + 6 * bq[b] + c];
  } else if (i < nb36 + n) {
    packed[i] = H[(long)npad * ld + (i - nb36)];
  } else if (i == nb36 + n) {
    packed[i] = meta[2] ? 1.0 : 0.0;
  } else if (i == nb36 + n + 1) {
    packed[i] = host_flags ? 1.0 : 0.0;
  }
}

__global__ __launch_bounds__(256) void ba_unpack_blocks_kernel(
    double* __restrict__ H, int ld, int npad, int n, const int* __restrict__ bp, const int* __restrict__ bq,
    int n_blocks, int* __restrict__ meta, const double* __restrict__ packed) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long nb36 = (long)n_blocks * 36;
  if (i < nb36) {
    const int b = (int)(i / 36), rc = (int)(i - (long)b * 36), r = rc / 6, c = rc - 6 * r;
    H[(long)(6 * bp[b] + r) * ld + 6 * bq[b] + c] = packed[i];
  } else if (i < nb36 + n) {
    H[(long)npad * ld + (i - nb36)] = packed[i];
  } else if (i == nb36 + n) {
    if (packed[nb36 + n] != 0.0 || packed[nb36 + n + 1] != 0.0) atomicOr(&meta[2], 4);   // some rank flagged: nobody updates
  }
}

// dense exchange: the flags travel in a separate 2-element buffer
__global__ void ba_flags_get_kernel(const int* __restrict__ meta, int host_flags, double* __restrict__ status) {
  status[0] = meta[2] ? 1.0 : 0.0; status[1] = host_flags ? 1.0 : 0.0;
}
__global__ void ba_flags_set_kernel(int* __restrict__ meta, const double* __restrict__ status) {
  if (status[0] != 0.0 || status[1] != 0.0) atomicOr(&meta[2], 4);
}

// ------------------------------------------------------------------------------------------ host
int check_args(int F, int E, int ht, int wd, int t0, int t1) {
  if (F <= 0 || E < 0 || ht <= 0 || wd <= 0) return DH_ERR_ARG;
  if (t0 < 0 || t1 < t0 || t1 > F) return DH_ERR_ARG;
  return DH_OK;
}

int run_prep(const BaLayout& L, char* ws, const int64_t* ii, const int64_t* jj, int E, int F, int t0, int t1, int n_eta_rows,
             hipStream_t st) {
  hipLaunchKernelGGL(ba_prep_kernel, dim3(1), dim3(1024), 0, st, ii, jj, E, F, t0, t1, n_eta_rows,
                     (int*)(ws + L.kmap), (int*)(ws + L.kx), (int*)(ws + L.eoff), (int*)(ws + L.cursor),
                     (int*)(ws + L.eidx), (int*)(ws + L.meta));
  DH_LAUNCH_CHECK();
  return DH_OK;
}

int run_build(const BaLayout& L, char* ws, const float* poses, const float* disps, const float* intr,
              const float* disps_sens, const float* targets, const float* weights, const float* eta,
              int n_eta_rows, const float* alpha, const int64_t* ii, const int64_t* jj, int F, int E, int HW, int wd,
              int t0, int motion_only, hipStream_t st) {
  double* H = (double*)(ws + L.H);
  if (hipMemsetAsync(H, 0, sizeof(double) * (size_t)(L.npad + NB) * L.ld, st) != hipSuccess) return DH_ERR_LAUNCH;
  const int* kx = (const int*)(ws + L.kx); const int* eoff = (const int*)(ws + L.eoff);
  const int* eidx = (const int*)(ws + L.eidx); const int* meta = (const int*)(ws + L.meta);
  dim3 grid(F, L.NS);
  if (motion_only)
    hipLaunchKernelGGL(ba_build_kernel<true>, grid, dim3(256), 0, st, poses, disps, intr, disps_sens, targets,
                       weights, eta, n_eta_rows, alpha, jj, kx, eoff, eidx, meta, HW, wd, L.NS,
                       (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)(ws + L.Hpart));
  else
    hipLaunchKernelGGL(ba_build_kernel<false>, grid, dim3(256), 0, st, poses, disps, intr, disps_sens, targets,
                       weights, eta, n_eta_rows, alpha, jj, kx, eoff, eidx, meta, HW, wd, L.NS,
                       (float*)(ws + L.Q), (float*)(ws + L.W), (float*)(ws + L.Ei), (float*)(ws + L.Ej),
                       (float*)(ws + L.Hpart));
  DH_LAUNCH_CHECK();
  if (E > 0 && L.P > 0) {
    hipLaunchKernelGGL(ba_pose_blocks_kernel, dim3(E), dim3(64), 0, st, poses, ii, jj,
                       (const float*)(ws + L.Hpart), E, F, L.NS, t0, L.P, H, L.ld, L.npad);
    DH_LAUNCH_CHECK();
  }
  if (!motion_only && L.P > 0) {
    const int kest = std::max(1, std::min(F, L.P + E));
    // ~8 workgroups per CU (4 resident) hide the HBM latency of the operand loads; every extra strip multiplies the fp64 atomics
    int NSG = std::max(1, std::min((2048 + kest - 1) / kest, std::max(1, HW / 256)));
    if (opts().gram_strips > 0) NSG = std::min(opts().gram_strips, std::max(1, HW / 64));
    hipLaunchKernelGGL(ba_gram_kernel<true>, dim3(F, NSG), dim3(256), 0, st, (const float*)(ws + L.Q),
                       (const float*)(ws + L.W), (const float*)(ws + L.Ei), (const float*)(ws + L.Ej), jj, kx,
                       eoff, eidx, meta, HW, NSG, t0, L.P, H, L.ld, L.npad);
    DH_LAUNCH_CHECK();
    if (E >= GS) {                                         // a second chunk takes a depth block with at least GS edges
      hipLaunchKernelGGL(ba_gram_kernel<false>, dim3(F, NSG), dim3(256), 0, st, (const float*)(ws + L.Q),
                         (const float*)(ws + L.W), (const float*)(ws + L.Ei), (const float*)(ws + L.Ej), jj, kx,
                         eoff, eidx, meta, HW, NSG, t0, L.P, H, L.ld, L.npad);
      DH_LAUNCH_CHECK();
    }
  }
  return DH_OK;
}

int run_finish(const BaLayout& L, char* ws, float* poses, float* disps, const int64_t* jj, int F, int HW,
               int t0, float lm, float ep, int motion_only, float* dx_out, float* dz_out, int n_dz_rows,
               hipStream_t st, int own_lo = 0, int own_hi = 1 << 30) {
  double* H = (double*)(ws + L.H);
  double* x = (double*)(ws + L.x);
  double* Linv = (double*)(ws + L.Linv);
  int* meta = (int*)(ws + L.meta);
  float* dxw = (float*)(ws + L.dx);
  if (L.P > 0) {
    int* flow = (int*)(ws + L.flow);
    hipLaunchKernelGGL(ba_damp_kernel, dim3((L.npad + 255) / 256), dim3(256), 0, st, H, L.ld, L.n, L.npad,
                       (double)lm, (double)ep, meta, flow, FLOW_HDR + (L.nbk + 1) * L.nbk);
    const int nbrows = L.nbk + 1;                        // + rhs block row
    DH_LDS_OPTIN(&chol_panel_kernel, 72 * 1024);
    DH_LDS_OPTIN(&chol_inverse_kernel, 80 * 1024);
    DH_LDS_OPTIN(&chol_update_kernel, 72 * 1024);
    DH_LDS_OPTIN(&chol_step_kernel<true>, 72 * 1024);
    DH_LDS_OPTIN(&chol_step_kernel<false>, 72 * 1024);
    DH_LDS_OPTIN(&chol_backsub_kernel, 112 * 1024);
    const size_t lds_panel = sizeof(double) * 2 * NB * LDB;
    const size_t lds_inv = sizeof(double) * (2 * NB * LDB + 3 * SB * (SB + 1) + NB);
    const size_t lds_upd = sizeof(double) * 2 * NB * LDB;
    const bool two_launch = opts().chol_lookahead == 0;
    double* Ldiag = (double*)(ws + L.Ldiag);
#ifdef DH_ABLATION
    if (opts().chol_lookahead >= 2) {                    // dataflow schedule: one persistent launch (A/B builds: measured 3.67 vs 3.51 ms per global BA)
      const int ntasks = L.nbk * (L.nbk + 1) / 2;
      const dim3 grid(std::min(ntasks, 2 * device_cus()));
      if (opts().chol_lookahead == 3) {                  // second form (LDS-DMA operands, grouped acquires)
        DH_LDS_OPTIN(&chol_flow_dma_kernel, 140 * 1024);
        hipLaunchKernelGGL(chol_flow_dma_kernel, grid, dim3(256), 2 * lds_panel + 256 * sizeof(double) + 16, st, H, L.ld, L.nbk, meta, Ldiag, flow, 0);
      } else {
        DH_LDS_OPTIN(&chol_flow_kernel, 72 * 1024);
        hipLaunchKernelGGL(chol_flow_kernel, grid, dim3(256), lds_panel + 256 * sizeof(double) + 16, st, H, L.ld, L.nbk, meta, Ldiag, flow, 0);
      }
      DH_LAUNCH_CHECK();
    } else
#endif
    if (two_launch) {                                    // reference schedule: panel, then the whole trailing update
      for (int j = 0; j < L.nbk; ++j) {
        const int m = nbrows - j - 1;
        hipLaunchKernelGGL(chol_panel_kernel, dim3(m), dim3(256), lds_panel, st, H, L.ld, j, meta, Ldiag);
        DH_LAUNCH_CHECK();
        if (m > 1) hipLaunchKernelGGL(chol_update_kernel, dim3(m - 1, m), dim3(256), lds_upd, st, H, L.ld, j, L.nbk);
        DH_LAUNCH_CHECK();
      }
    } else {
      hipLaunchKernelGGL(chol_panel_kernel, dim3(nbrows - 1), dim3(256), lds_panel, st, H, L.ld, 0, meta, Ldiag);
      DH_LAUNCH_CHECK();
      for (int j = 0; j + 1 < L.nbk; ++j) {
        const int nP = nbrows - j - 2;                   // block rows below the diagonal block of column j+1
        if (opts().chol_regpanel)
          hipLaunchKernelGGL(chol_step_kernel<true>, dim3(nP + nP * (nP + 1) / 2), dim3(256), lds_panel + 256 * sizeof(double), st, H, L.ld, j, L.nbk, nP, meta, Ldiag);
        else
          hipLaunchKernelGGL(chol_step_kernel<false>, dim3(nP + nP * (nP + 1) / 2), dim3(256), lds_panel, st, H, L.ld, j, L.nbk, nP, meta, Ldiag);
        DH_LAUNCH_CHECK();
      }
    }
    hipLaunchKernelGGL(chol_inverse_kernel, dim3(L.nbk), dim3(256), lds_inv, st, (const double*)Ldiag, Linv);
    DH_LAUNCH_CHECK();
    for (int jhi = L.nbk - 1; jhi >= 0; jhi -= BG)
      hipLaunchKernelGGL(chol_backsub_kernel, dim3(std::max(1, jhi - BG + 2)), dim3(256),
                         sizeof(double) * (BG + 1) * NB * LDB, st, H, L.ld, jhi, L.npad, Linv, x);
    hipLaunchKernelGGL(ba_dx_kernel, dim3((L.n + 255) / 256), dim3(256), 0, st, x, meta, L.n, dxw, dx_out);
    DH_LAUNCH_CHECK();
  }
  if (!motion_only) {
    hipLaunchKernelGGL(ba_backsub_kernel, dim3(F, (HW + 255) / 256), dim3(256), 0, st, disps,
                       (const float*)(ws + L.Q), (const float*)(ws + L.W), (const float*)(ws + L.Ei),
                       (const float*)(ws + L.Ej), dxw, jj, (const int*)(ws + L.kx), (const int*)(ws + L.eoff),
                       (const int*)(ws + L.eidx), meta, HW, t0, L.P, dz_out, n_dz_rows, own_lo, own_hi);
    DH_LAUNCH_CHECK();
  }
  if (L.P > 0) {
    hipLaunchKernelGGL(ba_pose_retr_kernel, dim3((L.P + 63) / 64), dim3(64), 0, st, poses, dxw, t0, L.P);
    DH_LAUNCH_CHECK();
  }
  return DH_OK;
}

}  // namespace

namespace {
// default (DH_BA_STRICT=1): flagged arguments (meta[2]: an edge outside the frame buffer, eta without one row per depth block) are
// reported as DH_ERR_ARG.  The flagged call has applied no update either way.
// Round 6: the flags are final after the call's FIRST kernel (ba_prep_kernel is the only single-process writer of meta[2]), so they are
// copied to pinned host memory right behind it (strict_arm) and the call ends by waiting for THAT copy (an event), not for the stream:
// the host gets its answer ~0.1 ms into the call's 3.5 ms of device work and enqueues the caller's next launches under the solve.  With the
// stream synchronisation at the end (rounds 1-5) the device idled 0.33 ms per update iteration while the launch queue refilled
// (profiles/r06_y_step_timeline.txt).  One probe per host thread; a call that was not armed falls back to the synchronisation.
struct StrictProbe { int* host = nullptr; hipEvent_t ev = nullptr; bool armed = false; };
thread_local StrictProbe t_probe;

void strict_arm(const BaLayout& L, char* ws, hipStream_t st) {
  StrictProbe& p = t_probe;
  p.armed = false;
  if (opts().ba_strict != 1) return;        // 0: no check at all; 2: rounds 1-5, the check behind a stream synchronisation (A/B runs)
  if (!p.host) {
    if (hipHostMalloc(reinterpret_cast<void**>(&p.host), 64, hipHostMallocDefault) != hipSuccess) { p.host = nullptr; return; }
    if (hipEventCreateWithFlags(&p.ev, hipEventDisableTiming) != hipSuccess) { (void)hipHostFree(p.host); p.host = nullptr; p.ev = nullptr; return; }
  }
  if (hipMemcpyAsync(p.host, ws + L.meta, 4 * sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess) return;
  if (hipEventRecord(p.ev, st) != hipSuccess) return;
  p.armed = true;
}

int strict_check(const BaLayout& L, char* ws, hipStream_t st) {
  if (!opts().ba_strict) return DH_OK;
  StrictProbe& p = t_probe;
  if (p.armed) {
    p.armed = false;
    if (hipEventSynchronize(p.ev) != hipSuccess) return DH_ERR_LAUNCH;
    return p.host[2] ? DH_ERR_ARG : DH_OK;
  }
  int flags[4] = {0, 0, 0, 0};
  if (hipMemcpyAsync(flags, ws + L.meta, sizeof(flags), hipMemcpyDeviceToHost, st) != hipSuccess) return DH_ERR_LAUNCH;
  if (hipStreamSynchronize(st) != hipSuccess) return DH_ERR_LAUNCH;
  return flags[2] ? DH_ERR_ARG : DH_OK;
}
}  // namespace

extern "C" size_t dh_ba_workspace_bytes(int num_frames, int n_edges, int ht, int wd, int t0, int t1, int motion_only) {
  if (check_args(num_frames, n_edges, ht, wd, t0, t1) != DH_OK) return 0;
  return make_layout(num_frames, n_edges, ht * wd, t0, t1, motion_only).total;
}

extern "C" int dh_ba_system_shape(int t0, int t1, int* rows, int* cols) {
  if (t0 < 0 || t1 < t0 || !rows || !cols) return DH_ERR_ARG;
  const int n = 6 * (t1 - t0), npad = (n + NB - 1) / NB * NB;
  *rows = npad + NB; *cols = npad;
  return DH_OK;
}

static int ba_build_impl(int strict, const float* poses, const float* disps, const float* intrinsics, const float* disps_sens,
                         const float* alpha, const float* targets, const float* weights, const float* eta,
                           const int64_t* ii, const int64_t* jj,
                           int num_frames, int n_edges, int n_eta_rows, int ht, int wd,
                           int t0, int t1, int motion_only,
                           double** Hsys_out, double** bsys_out, void* workspace, size_t workspace_bytes,
                           dh_stream_t stream) {
  int rc = check_args(num_frames, n_edges, ht, wd, t0, t1);
  if (rc != DH_OK) return rc;
  if (!poses || !disps || !intrinsics || !workspace) return DH_ERR_ARG;
  if (n_edges > 0 && (!targets || !weights || !ii || !jj)) return DH_ERR_ARG;
  if (!motion_only && (!disps_sens || (n_eta_rows > 0 && !eta))) return DH_ERR_ARG;
  const BaLayout L = make_layout(num_frames, n_edges, ht * wd, t0, t1, motion_only);
  if (workspace_bytes < L.total || ((uintptr_t)workspace & 255)) return DH_ERR_WORKSPACE;   // 16-byte operand loads, fp64 blocks
  char* ws = (char*)workspace;
  hipStream_t st = (hipStream_t)stream;
  rc = run_prep(L, ws, ii, jj, n_edges, num_frames, t0, t1, motion_only ? -1 : n_eta_rows, st);
  if (rc != DH_OK) return rc;
  if (strict) strict_arm(L, ws, st);
  rc = run_build(L, ws, poses, disps, intrinsics, disps_sens, targets, weights, eta, n_eta_rows, alpha, ii, jj,
                 num_frames, n_edges, ht * wd, wd, t0, motion_only, st);
  if (Hsys_out) *Hsys_out = (double*)(ws + L.H);
  if (bsys_out) *bsys_out = (double*)(ws + L.H) + (size_t)L.npad * L.ld;
  if (rc != DH_OK) return rc;
  return strict ? strict_check(L, ws, st) : DH_OK;
}

extern "C" int dh_ba_build(const float* poses, const float* disps, const float* intrinsics, const float* disps_sens,
                           const float* targets, const float* weights, const float* eta,
                           const int64_t* ii, const int64_t* jj,
                           int num_frames, int n_edges, int n_eta_rows, int ht, int wd,
                           int t0, int t1, int motion_only,
                           double** Hsys_out, double** bsys_out, void* workspace, size_t workspace_bytes,
                           dh_stream_t stream) {
  return ba_build_impl(1, poses, disps, intrinsics, disps_sens, nullptr, targets, weights, eta, ii, jj, num_frames, n_edges, n_eta_rows,
                       ht, wd, t0, t1, motion_only, Hsys_out, bsys_out, workspace, workspace_bytes, stream);
}

// the edge-sharded solver's build: never synchronises -- a rank-local error before the collective would leave the other
// ranks waiting in it; the argument flag travels with the exchanged buffer instead (dh_ba_pack_blocks / dh_ba_exchange_flags)
extern "C" int dh_ba_build_shard(const float* poses, const float* disps, const float* intrinsics, const float* disps_sens,
                                 const float* targets, const float* weights, const float* eta,
                                 const int64_t* ii, const int64_t* jj,
                                 int num_frames, int n_edges, int n_eta_rows, int ht, int wd,
                                 int t0, int t1, int motion_only,
                                 double** Hsys_out, double** bsys_out, void* workspace, size_t workspace_bytes,
                                 dh_stream_t stream) {
  return ba_build_impl(0, poses, disps, intrinsics, disps_sens, nullptr, targets, weights, eta, ii, jj, num_frames, n_edges, n_eta_rows,
                       ht, wd, t0, t1, motion_only, Hsys_out, bsys_out, workspace, workspace_bytes, stream);
}

// dh_ba_build_shard with the per-pixel weight of the sensor-depth prior of dh_ba_ex (BASELINE configs[4] is defined on the sharded
// path): alpha [num_frames,ht,wd] f32 or NULL (= the reference's constant 0.05).  A depth block is assembled from the edges of its
// source frame, which all live on the frame's owner: the prior enters the system once, on that rank, like in dh_ba_ex.
extern "C" int dh_ba_build_shard_ex(const float* poses, const float* disps, const float* intrinsics, const float* disps_sens,
                                    const float* alpha, const float* targets, const float* weights, const float* eta,
                                    const int64_t* ii, const int64_t* jj,
                                    int num_frames, int n_edges, int n_eta_rows, int ht, int wd,
                                    int t0, int t1, int motion_only,
                                    double** Hsys_out, double** bsys_out, void* workspace, size_t workspace_bytes,
                                    dh_stream_t stream) {
  return ba_build_impl(0, poses, disps, intrinsics, disps_sens, alpha, targets, weights, eta, ii, jj, num_frames, n_edges, n_eta_rows,
                       ht, wd, t0, t1, motion_only, Hsys_out, bsys_out, workspace, workspace_bytes, stream);
}

extern "C" size_t dh_ba_packed_len(int n_blocks, int t0, int t1) {
  if (n_blocks < 0 || t0 < 0 || t1 < t0) return 0;
  return (size_t)n_blocks * 36 + 6 * (size_t)(t1 - t0) + 2;
}

static int pack_args(int num_frames, int n_edges, int ht, int wd, int t0, int t1, int motion_only, const void* workspace,
                     size_t workspace_bytes, const int32_t* bp, const int32_t* bq, int n_blocks, const void* packed, BaLayout* L) {
  int rc = check_args(num_frames, n_edges, ht, wd, t0, t1);
  if (rc != DH_OK) return rc;
  if (!workspace || !packed || n_blocks < 0 || (n_blocks > 0 && (!bp || !bq))) return DH_ERR_ARG;
  *L = make_layout(num_frames, n_edges, ht * wd, t0, t1, motion_only);
  if (workspace_bytes < L->total || ((uintptr_t)workspace & 255)) return DH_ERR_WORKSPACE;
  return DH_OK;
}

extern "C" int dh_ba_pack_blocks(const void* workspace, size_t workspace_bytes, int num_frames, int n_edges, int ht, int wd,
                                 int t0, int t1, int motion_only, const int32_t* bp, const int32_t* bq, int n_blocks,
                                 int host_flags, double* packed, dh_stream_t stream) {
  BaLayout L;
  int rc = pack_args(num_frames, n_edges, ht, wd, t0, t1, motion_only, workspace, workspace_bytes, bp, bq, n_blocks, packed, &L);
  if (rc != DH_OK) return rc;
  const char* ws = (const char*)workspace;
  const long total = (long)n_blocks * 36 + L.n + 2;
  hipLaunchKernelGGL(ba_pack_blocks_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const double*)(ws + L.H), L.ld, L.npad, L.n, bp, bq, n_blocks, (const int*)(ws + L.meta), host_flags, packed);
  DH_LAUNCH_CHECK();
  return DH_OK;
}

extern "C" int dh_ba_unpack_blocks(void* workspace, size_t workspace_bytes, int num_frames, int n_edges, int ht, int wd,
                                   int t0, int t1, int motion_only, const int32_t* bp, const int32_t* bq, int n_blocks,
                                   const double* packed, dh_stream_t stream) {
  BaLayout L;
  int rc = pack_args(num_frames, n_edges, ht, wd, t0, t1, motion_only, workspace, workspace_bytes, bp, bq, n_blocks, packed, &L);
  if (rc != DH_OK) return rc;
  char* ws = (char*)workspace;
  const long total = (long)n_blocks * 36 + L.n + 1;
  hipLaunchKernelGGL(ba_unpack_blocks_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (double*)(ws + L.H), L.ld, L.npad, L.n, bp, bq, n_blocks, (int*)(ws + L.meta), packed);
  DH_LAUNCH_CHECK();
  return DH_OK;
}

// dense exchange (no block pattern): get -> all-reduce(SUM) of status[2] -> set
extern "C" int dh_ba_exchange_flags(void* workspace, size_t workspace_bytes, int num_frames, int n_edges, int ht, int wd,
                                    int t0, int t1, int motion_only, int host_flags, double* status, int set, dh_stream_t stream) {
  BaLayout L;
  int rc = pack_args(num_frames, n_edges, ht, wd, t0, t1, motion_only, workspace, workspace_bytes, nullptr, nullptr, 0, status, &L);
  if (rc != DH_OK) return rc;
  char* ws = (char*)workspace;
  if (set) hipLaunchKernelGGL(ba_flags_set_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (int*)(ws + L.meta), (const double*)status);
  else hipLaunchKernelGGL(ba_flags_get_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (const int*)(ws + L.meta), host_flags, status);
  DH_LAUNCH_CHECK();
  return DH_OK;
}

static int ba_finish_impl(float* poses, float* disps, const int64_t* jj,
                          int num_frames, int n_edges, int ht, int wd, int t0, int t1,
                          float lm, float ep, int motion_only, float* dx_out, float* dz_out,
                          void* workspace, size_t workspace_bytes, dh_stream_t stream, int own_lo, int own_hi) {
  int rc = check_args(num_frames, n_edges, ht, wd, t0, t1);
  if (rc != DH_OK) return rc;
  if (!poses || !disps || !workspace) return DH_ERR_ARG;
  const BaLayout L = make_layout(num_frames, n_edges, ht * wd, t0, t1, motion_only);
  if (workspace_bytes < L.total || ((uintptr_t)workspace & 255)) return DH_ERR_WORKSPACE;   // 16-byte operand loads, fp64 blocks
  if (n_edges > 0 && !jj) return DH_ERR_ARG;
  return run_finish(L, (char*)workspace, poses, disps, jj, num_frames, ht * wd, t0, lm, ep, motion_only,
                    dx_out, dz_out, num_frames, (hipStream_t)stream, own_lo, own_hi);
}

// the depth blocks behind a built system (after dh_ba_build): Qinv = 1 / C and w, [K, ht*wd] f32 each, rows in the order of
// kx [K] (the sorted source frames) -- the reference's `C` and `w` of ba_cuda (droid_kernels.cu:1407-1408); read by the parity
// tests, and by callers that want per-pixel depth covariances (Q is the marginal's diagonal)
extern "C" int dh_ba_depth_blocks(const void* workspace, size_t workspace_bytes, int num_frames, int n_edges, int ht, int wd,
                                  int t0, int t1, const float** Qinv_out, const float** w_out, const int** kx_out, const int** K_out) {
  int rc = check_args(num_frames, n_edges, ht, wd, t0, t1);
  if (rc != DH_OK) return rc;
  if (!workspace) return DH_ERR_ARG;
  const BaLayout L = make_layout(num_frames, n_edges, ht * wd, t0, t1, 0);
  if (workspace_bytes < L.total || ((uintptr_t)workspace & 255)) return DH_ERR_WORKSPACE;
  const char* ws = (const char*)workspace;
  if (Qinv_out) *Qinv_out = (const float*)(ws + L.Q);
  if (w_out) *w_out = (const float*)(ws + L.W);
  if (kx_out) *kx_out = (const int*)(ws + L.kx);
  if (K_out) *K_out = (const int*)(ws + L.meta);          // device int: the number of depth blocks
  return DH_OK;
}

extern "C" int dh_ba_finish(float* poses, float* disps, const int64_t* jj,
                            int num_frames, int n_edges, int ht, int wd, int t0, int t1,
                            float lm, float ep, int motion_only, float* dx_out, float* dz_out,
                            void* workspace, size_t workspace_bytes, dh_stream_t stream) {
  return ba_finish_impl(poses, disps, jj, num_frames, n_edges, ht, wd, t0, t1, lm, ep, motion_only, dx_out, dz_out,
                        workspace, workspace_bytes, stream, 0, 1 << 30);
}

// dh_ba_finish for a rank that owns the depth maps of frames [own_lo, own_hi): the depths (and dz rows) of every other
// frame stay untouched; poses are replicated and retracted by every rank
extern "C" int dh_ba_finish_owned(float* poses, float* disps, const int64_t* jj,
                                  int num_frames, int n_edges, int ht, int wd, int t0, int t1,
                                  float lm, float ep, int motion_only, int own_lo, int own_hi, float* dx_out, float* dz_out,
                                  void* workspace, size_t workspace_bytes, dh_stream_t stream) {
  if (own_lo > own_hi) return DH_ERR_ARG;
  return ba_finish_impl(poses, disps, jj, num_frames, n_edges, ht, wd, t0, t1, lm, ep, motion_only, dx_out, dz_out,
                        workspace, workspace_bytes, stream, own_lo, own_hi);
}

extern "C" int dh_ba_ex(float* poses, float* disps, const float* intrinsics, const float* disps_sens, const float* alpha,
                        const float* targets, const float* weights, const float* eta,
                        const int64_t* ii, const int64_t* jj,
                        int num_frames, int n_edges, int n_eta_rows, int ht, int wd,
                        int t0, int t1, int iterations, float lm, float ep, int motion_only,
                        float* dx_out, float* dz_out, void* workspace, size_t workspace_bytes, dh_stream_t stream) {
  int rc = check_args(num_frames, n_edges, ht, wd, t0, t1);
  if (rc != DH_OK) return rc;
  if (iterations < 0) return DH_ERR_ARG;
  if (!poses || !disps || !intrinsics || !workspace) return DH_ERR_ARG;
  if (n_edges > 0 && (!targets || !weights || !ii || !jj)) return DH_ERR_ARG;
  if (!motion_only && (!disps_sens || (n_eta_rows > 0 && !eta))) return DH_ERR_ARG;
  const int HW = ht * wd;
  const BaLayout L = make_layout(num_frames, n_edges, HW, t0, t1, motion_only);
  if (workspace_bytes < L.total || ((uintptr_t)workspace & 255)) return DH_ERR_WORKSPACE;   // 16-byte operand loads, fp64 blocks
  char* ws = (char*)workspace;
  hipStream_t st = (hipStream_t)stream;
  rc = run_prep(L, ws, ii, jj, n_edges, num_frames, t0, t1, motion_only ? -1 : n_eta_rows, st);
  if (rc != DH_OK) return rc;
  strict_arm(L, ws, st);
  for (int it = 0; it < iterations; ++it) {
    rc = run_build(L, ws, poses, disps, intrinsics, disps_sens, targets, weights, eta, n_eta_rows, alpha, ii, jj,
                   num_frames, n_edges, HW, wd, t0, motion_only, st);
    if (rc != DH_OK) return rc;
    const bool last = it == iterations - 1;
    rc = run_finish(L, ws, poses, disps, jj, num_frames, HW, t0, lm, ep, motion_only,
                    last ? dx_out : nullptr, last ? dz_out : nullptr, n_eta_rows, st);
    if (rc != DH_OK) return rc;
  }
  return strict_check(L, ws, st);
}

extern "C" int dh_ba(float* poses, float* disps, const float* intrinsics, const float* disps_sens,
                     const float* targets, const float* weights, const float* eta,
                     const int64_t* ii, const int64_t* jj,
                     int num_frames, int n_edges, int n_eta_rows, int ht, int wd,
                     int t0, int t1, int iterations, float lm, float ep, int motion_only,
                     float* dx_out, float* dz_out, void* workspace, size_t workspace_bytes, dh_stream_t stream) {
  return dh_ba_ex(poses, disps, intrinsics, disps_sens, nullptr, targets, weights, eta, ii, jj, num_frames, n_edges, n_eta_rows, ht, wd,
                  t0, t1, iterations, lm, ep, motion_only, dx_out, dz_out, workspace, workspace_bytes, stream);
}
