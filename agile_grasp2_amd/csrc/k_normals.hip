// k_normals.hip -- K1: per-point PCA plane normals.
//
// Replaces HandSearch::calculateNormalsOMP (src/agile_grasp2/hand_search.cpp:83-94), i.e.
// pcl::NormalEstimationOMP with radius 0.01 and viewpoint (0,0,0): neighbours within the radius
// (self included, strict float d2 < (float)(r*r)); fewer than 3 => NaN; raw moments accumulated in
// float in canonical (sorted-position) order; covariance E[xx^T] - mm^T; eigenvector of the smallest
// eigenvalue (Jacobi in f64); flipped towards the viewpoint.
//
// One thread per point in SORTED order: the 64 lanes of a wave are spatial neighbours, so their
// 3x3 row spans overlap and the float4 point loads are served from L1/L2.  HBM-bound by design:
// algorithmic bytes = N * (K1 * 12 + 12).
#include "ag2_internal.h"

namespace ag2 {

__global__ void __launch_bounds__(256) k_normals(const float4* __restrict__ pts,
                                                 const unsigned* __restrict__ cell, GridDesc g_arg,
                                                 const GridDesc* __restrict__ gp,
                                                 float r2f, float rq, float4* __restrict__ nrm,
                                                 DevStats* st) {
  // gp (frame mode): grid description left in device memory by k_grid_desc; the launch covers the
  // frame's maximum point count and the threads beyond n_valid leave
  const GridDesc g = gp ? *gp : g_arg;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  int cnt = 0;
  if (i < g.n_valid) {
    const float4 q = pts[i];
    int lo[3], hi[3];
    const float qq[3] = {q.x, q.y, q.z};
#pragma unroll
    for (int a = 0; a < 3; a++) {
      lo[a] = max(cell_of(qq[a] - rq, g.o[a], g.inv), 0);
      hi[a] = min(cell_of(qq[a] + rq, g.o[a], g.inv), g.dims[a] - 1);
    }
    float a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0, a8 = 0;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 A01 = {0.f, 0.f}, A23 = {0.f, 0.f}, A45 = {0.f, 0.f}, A67 = {0.f, 0.f};  // the row walk below
    const f2 qxy = {q.x, q.y};
    // The kernel is bound by the latency of each thread's chain of dependent loads (one wave per
    // SIMD slot, every wave runs once), so the chain is kept short: the span bounds of ALL stencil
    // rows (<= 4 x 4: q +- 1.001 r reaches 3, rarely 4, cells per axis) are requested first, and a
    // row is walked four points per step with the four loads issued together.  The moments are
    // still accumulated one point at a time in canonical order (cz, cy, then sorted position).
    constexpr int kRows = 16;
    int nrows = (hi[1] - lo[1] + 1) * (hi[2] - lo[2] + 1);
    if (nrows > kRows) {  // normals_radius much larger than grid_cell: plain nested walk, same order
      for (int cz = lo[2]; cz <= hi[2]; cz++)
        for (int cy = lo[1]; cy <= hi[1]; cy++) {
          const int rowbase = (cz * g.dims[1] + cy) * g.dims[0];
          const int e = (int)cell[rowbase + hi[0] + 1];
          for (int j = (int)cell[rowbase + lo[0]]; j < e; j++) {
            const float4 p = pts[j];
            const float dx = p.x - q.x, dy = p.y - q.y, dz = p.z - q.z;
            const float d2 = (dx * dx + dy * dy) + dz * dz;
            if (d2 < r2f) {
              a0 = a0 + p.x * p.x;
              a1 = a1 + p.x * p.y;
              a2 = a2 + p.x * p.z;
              a3 = a3 + p.y * p.y;
              a4 = a4 + p.y * p.z;
              a5 = a5 + p.z * p.z;
              a6 = a6 + p.x;
              a7 = a7 + p.y;
              a8 = a8 + p.z;
              cnt++;
            }
          }
        }
      nrows = 0;
    }
    int rb[kRows], re[kRows];
    {
      // The kernel's time is the VALU work of the distance tests, six of seven of which fail: each
      // row's x-cell range is shrunk to what the sphere |p - q| < r can reach given the row's
      // distance in y and z (a row out of reach is skipped).  Purely conservative -- mg of slack on
      // every bound, two orders above the float rounding of these few operations -- so the exact
      // per-point test still decides and the accepted set and its order are unchanged.
      const float mg = 1.0e-4f, h = 1.0f / g.inv, rm = rq + mg;
      int cy = lo[1], cz = lo[2];
#pragma unroll
      for (int r = 0; r < kRows; r++) {
        rb[r] = 0;
        re[r] = 0;
        if (r < nrows) {
          const float dyl = (g.o[1] + (float)cy * h) - q.y - mg, dyh = dyl + h + 2.0f * mg;
          const float dzl = (g.o[2] + (float)cz * h) - q.z - mg, dzh = dzl + h + 2.0f * mg;
          const float dym = dyl > 0.f ? dyl : (dyh < 0.f ? -dyh : 0.f);
          const float dzm = dzl > 0.f ? dzl : (dzh < 0.f ? -dzh : 0.f);
          const float rho2 = rm * rm - dym * dym - dzm * dzm;
          if (rho2 > 0.f) {
            const float rho = __builtin_sqrtf(rho2) + mg;
            const int cxa = max(lo[0], cell_of(q.x - rho, g.o[0], g.inv));
            const int cxb = min(hi[0], cell_of(q.x + rho, g.o[0], g.inv));
            if (cxa <= cxb) {
              const int rowbase = (cz * g.dims[1] + cy) * g.dims[0];
              rb[r] = (int)cell[rowbase + cxa];
              re[r] = (int)cell[rowbase + cxb + 1];
            }
          }
          if (++cy > hi[1]) {
            cy = lo[1];
            cz++;
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < kRows; r++) {
      if (r < nrows) {
        const int e = re[r];
        for (int j = rb[r]; j < e; j += 4) {
          float4 p4[4];
#pragma unroll
          for (int k = 0; k < 4; k++) p4[k] = pts[min(j + k, e - 1)];
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const float4 p = p4[k];
            // two-wide packed f32 operations (v_pk_mul_f32 / v_pk_add_f32): every component is the
            // same IEEE multiply or add as before -- nothing is contracted or reassociated -- at half
            // the instruction count
            const f2 pxy = {p.x, p.y}, dxy = pxy - qxy;
            const float dz = p.z - q.z;
            const f2 dd = dxy * dxy;
            const float d2 = (dd.x + dd.y) + dz * dz;
            if (j + k < e && d2 < r2f) {
              const f2 pxx = {p.x, p.x}, pzy = {p.z, p.y}, pyz = {p.y, p.z}, pzz = {p.z, p.z};
              A01 = A01 + pxx * pxy;   // xx, xy
              A23 = A23 + pxy * pzy;   // xz, yy
              A45 = A45 + pyz * pzz;   // yz, zz
              A67 = A67 + pxy;         // x, y
              a8 = a8 + p.z;
              cnt++;
            }
          }
        }
      }
    }
    a0 = a0 + A01.x; a1 = a1 + A01.y; a2 = a2 + A23.x; a3 = a3 + A23.y;  // (one of the two sets is zero:
    a4 = a4 + A45.x; a5 = a5 + A45.y; a6 = a6 + A67.x; a7 = a7 + A67.y;  //  x + 0 is exact)
    float4 out;
    if (cnt < 3) {
      const float nanv = __builtin_nanf("");
      out = make_float4(nanv, nanv, nanv, 0.f);
    } else {
      const float fc = (float)cnt;
      a0 = a0 / fc; a1 = a1 / fc; a2 = a2 / fc; a3 = a3 / fc; a4 = a4 / fc;
      a5 = a5 / fc; a6 = a6 / fc; a7 = a7 / fc; a8 = a8 / fc;
      Sym3 m;
      m.a00 = (double)(a0 - a6 * a6);
      m.a01 = (double)(a1 - a6 * a7);
      m.a02 = (double)(a2 - a6 * a8);
      m.a11 = (double)(a3 - a7 * a7);
      m.a12 = (double)(a4 - a7 * a8);
      m.a22 = (double)(a5 - a8 * a8);
      const Eig3 e = jacobi3(m);
      const int mi = argmin3(e.d);
      const V3 v{mi == 0 ? e.v[0][0] : (mi == 1 ? e.v[0][1] : e.v[0][2]),
                 mi == 0 ? e.v[1][0] : (mi == 1 ? e.v[1][1] : e.v[1][2]),
                 mi == 0 ? e.v[2][0] : (mi == 1 ? e.v[2][1] : e.v[2][2])};
      const double nv = norm3(v);
      float fx = (float)(v.x / nv), fy = (float)(v.y / nv), fz = (float)(v.z / nv);
      const float vx = 0.0f - q.x, vy = 0.0f - q.y, vz = 0.0f - q.z;
      const float ct = (vx * fx + vy * fy) + vz * fz;
      if (ct < 0.0f) {
        fx = -fx; fy = -fy; fz = -fz;
      }
      out = make_float4(fx, fy, fz, 0.f);
    }
    nrm[i] = out;
  }
  const int tot = wave_sum_i(cnt);
  if (lane_id() == 0 && tot) atomicAdd(&st->sum_k1, (unsigned long long)tot);
}

int launch_normals(ag2_ctx* c) {
  if (!c->fm_on && c->n_valid == 0) return 0;
  const int nb = ((int)(c->fm_on ? c->fm_n_max : c->n_valid) + 255) / 256;
  hipLaunchKernelGGL(k_normals, dim3(nb), dim3(256), 0, c->stream, c->d_sorted.as<float4>(),
                     c->d_cell.as<unsigned>(), c->grid,
                     c->fm_on ? c->d_griddesc.as<GridDesc>() : (const GridDesc*)nullptr,
                     c->hc.r2_normals, c->hc.rq_normals,
                     c->d_nrm.as<float4>(), c->d_stats.as<DevStats>());
  AG2_HIP(c, hipGetLastError());
  return 0;
}

}  // namespace ag2
