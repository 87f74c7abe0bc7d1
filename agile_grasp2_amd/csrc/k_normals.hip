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

// ablation switches for tools/ab_build.sh (timing only, wrong results): 1 no Jacobi, 2 one stencil row
// only, 4 no statistics atomic
#ifndef AG2_EXP_NABL
#define AG2_EXP_NABL 0
#endif
namespace ag2 {

__global__ void __launch_bounds__(256) k_normals(const float4* __restrict__ pts,
                                                 const unsigned* __restrict__ cell, GridDesc g_arg,
                                                 const GridDesc* __restrict__ gp,
                                                 float r2f, float rq, float4* __restrict__ nrm,
                                                 DevStats* st) {
  // gp (frame mode): grid description left in device memory by k_grid_desc; the launch covers the
  // frame's maximum point count and the threads beyond n_valid leave
  const GridDesc g = gp ? *gp : g_arg;
  constexpr int kRows = 16;  // 4 x 4 stencil rows
  __shared__ int s_rb[kRows][256];             // [row][thread]: a thread's column is its own
  __shared__ unsigned short s_rn[kRows][256];  // span lengths (a span is three cells of one row)
  __shared__ int s_tot[4];
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
    // accumulators of the flat walk below, paired the way a float4 point sits in registers -- (x, y)
    // and (z, w) are the two aligned register pairs a 16-byte load returns, so no operand has to be
    // moved into place: (xx, xy), (xy, yy), (xz, yz), (zz, ww), (x, y), (z, w).  The second xy and
    // everything made from w are by-products nobody reads.
    f2 Ax = {0.f, 0.f}, Ay = {0.f, 0.f}, Az = {0.f, 0.f}, Aw = {0.f, 0.f}, Sxy = {0.f, 0.f}, Szw = {0.f, 0.f};
    const f2 qxy = {q.x, q.y}, qzw = {q.z, 0.f};
    // plain nested walk, same order: for a radius much larger than the cell (more than 4 x 4 stencil
    // rows) and for spans too long for the 16-bit lengths of the span table
    auto nested_walk = [&]() {
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
    };
    const int ny = hi[1] - lo[1] + 1, nz = hi[2] - lo[2] + 1;
    bool plain = ny > 4 || nz > 4;
    // Non-empty row spans of this thread, in canonical order (cz, then cy), in its own column of the
    // LDS tables.  The stencil is walked as a fixed 4 x 4 grid of (cy, cz) offsets (q +- 1.001 r
    // reaches 3, rarely 4, cells per axis; a wave skips the offsets none of its lanes has).
    // Six of seven distance tests fail, so each row's x-cell range is shrunk to what the sphere
    // |p - q| < r can reach given the row's distance in y and z (a row out of reach is skipped).
    // Purely conservative -- mg of slack on every bound, two orders above the rounding of these few
    // float operations (the square root may be the 1-ulp hardware one) -- so the exact per-point test
    // still decides and the accepted set and its order are unchanged.
    int nr = 0;
    if (!plain) {
      const float mg = 1.0e-4f, h = 1.0f / g.inv, rm = rq + mg;
      float dy2[4], dz2[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const float dyl = (g.o[1] + (float)(lo[1] + k) * h) - q.y - mg, dyh = dyl + h + 2.0f * mg;
        const float dzl = (g.o[2] + (float)(lo[2] + k) * h) - q.z - mg, dzh = dzl + h + 2.0f * mg;
        const float dym = dyl > 0.f ? dyl : (dyh < 0.f ? -dyh : 0.f);
        const float dzm = dzl > 0.f ? dzl : (dzh < 0.f ? -dzh : 0.f);
        dy2[k] = dym * dym;
        dz2[k] = dzm * dzm;
      }
      const float rm2 = rm * rm;
      const unsigned row0 = (unsigned)((lo[2] * g.dims[1] + lo[1]) * g.dims[0]);
      const unsigned zstep = (unsigned)(g.dims[1] * g.dims[0]), ystep = (unsigned)g.dims[0];
#pragma unroll
      for (int iz = 0; iz < 4; iz++)
#pragma unroll
        for (int iy = 0; iy < 4; iy++) {
          const float rho2 = rm2 - dy2[iy] - dz2[iz];
          if (iy < ny && iz < nz && rho2 > 0.f) {
            const float rho = __builtin_amdgcn_sqrtf(rho2) + mg;
            const int cxa = max(lo[0], cell_of(q.x - rho, g.o[0], g.inv));
            const int cxb = min(hi[0], cell_of(q.x + rho, g.o[0], g.inv));
            if (cxa <= cxb) {
              const unsigned rowbase = row0 + (unsigned)iz * zstep + (unsigned)iy * ystep;
              const int b = (int)cell[rowbase + (unsigned)cxa], e = (int)cell[rowbase + (unsigned)cxb + 1u];
              if (b < e) {
                plain = plain || (e - b > 65535);
                s_rb[nr][threadIdx.x] = b;
                s_rn[nr][threadIdx.x] = (unsigned short)(e - b);
                nr++;
              }
            }
          }
        }
    }
    if (plain) {
      nested_walk();
      nr = 0;
    }
#if AG2_EXP_NABL & 2
    nr = min(nr, 1);
#endif
    // One flat walk over the spans, four points per step, with the NEXT step's four loads in flight
    // while this step's points are tested and accumulated: a thread's chain of dependent load round
    // trips overlaps its arithmetic instead of adding to it.  A step that reaches past its span reads
    // the all-NaN point k_cell_sort leaves behind the sorted cloud: it fails the distance test like
    // any far point, no bounds test per candidate.
    if (nr > 0) {
      const unsigned sentinel = (unsigned)g.n_valid;
      int r = 0, j = s_rb[0][threadIdx.x], e = j + (int)s_rn[0][threadIdx.x];
      int nj = 0, ne = 0;  // bounds of row r + 1, read one row ahead
      if (nr > 1) {
        nj = s_rb[1][threadIdx.x];
        ne = nj + (int)s_rn[1][threadIdx.x];
      }
      // two register sets of four points, used alternately: no copies between the steps
      float4 pa[4], pb[4];
      auto fetch = [&](float4(&d)[4]) {
#pragma unroll
        for (int k = 0; k < 4; k++) d[k] = pts[j + k < e ? (unsigned)(j + k) : sentinel];
      };
      // moves on to the next step (the next span when this one is used up) and requests its points
      auto advance = [&](float4(&d)[4]) -> bool {
        j += 4;
        if (j >= e) {
          r++;
          if (r >= nr) return false;
          j = nj;
          e = ne;
          if (r + 1 < nr) {
            nj = s_rb[r + 1][threadIdx.x];
            ne = nj + (int)s_rn[r + 1][threadIdx.x];
          }
        }
        fetch(d);
        return true;
      };
      auto take = [&](const float4(&p4)[4]) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
          // two-wide packed f32 operations (v_pk_mul_f32 / v_pk_add_f32): every component is the
          // IEEE multiply or add of the scalar formulation -- nothing is contracted or reassociated --
          // at half the instruction count
          const f2 pxy = {p4[k].x, p4[k].y}, pzw = {p4[k].z, p4[k].w};
          const f2 dxy = pxy - qxy, dzw = pzw - qzw;
          const f2 dd = dxy * dxy, ee = dzw * dzw;
          const float d2 = (dd.x + dd.y) + ee.x;
          if (d2 < r2f) {
            const f2 bx = {pxy.x, pxy.x}, by = {pxy.y, pxy.y}, bz = {pzw.x, pzw.x};
            Ax = Ax + bx * pxy;    // xx, xy
            Ay = Ay + by * pxy;    // (xy), yy
            Az = Az + bz * pxy;    // xz, yz
            Aw = Aw + pzw * pzw;   // zz, (ww)
            Sxy = Sxy + pxy;       // x, y
            Szw = Szw + pzw;       // z, (w)
            cnt++;
          }
        }
      };
      fetch(pa);
      for (;;) {
        const bool mb = advance(pb);
        take(pa);
        if (!mb) break;
        const bool ma = advance(pa);
        take(pb);
        if (!ma) break;
      }
    }
    // the by-products count as used, so that the pairs stay pairs (and the loads 16 bytes wide)
    asm volatile("" ::"v"(Ay.x), "v"(Aw.y), "v"(Szw.y));
    a0 = a0 + Ax.x; a1 = a1 + Ax.y; a2 = a2 + Az.x; a3 = a3 + Ay.y;  // (one of the two sets is zero:
    a4 = a4 + Az.y; a5 = a5 + Aw.x; a6 = a6 + Sxy.x; a7 = a7 + Sxy.y;  //  x + 0 is exact)
    a8 = a8 + Szw.x;
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
#if AG2_EXP_NABL & 1
      Eig3 e; e.d[0]=m.a00; e.d[1]=m.a11; e.d[2]=m.a22; e.v[0][0]=m.a01; e.v[0][1]=m.a02; e.v[0][2]=m.a12; e.v[1][0]=1; e.v[1][1]=0; e.v[1][2]=0; e.v[2][0]=0; e.v[2][1]=1; e.v[2][2]=0;
#else
      const Eig3 e = jacobi3(m);
#endif
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
  // statistics: ONE atomic per workgroup.  (One per wave -- 4 700 on one address -- serialises in the
  // L2 at ~13 ns each: a 0.06 ms floor under the kernel, whatever else it does.)
  const int tot = wave_sum_i(cnt);
  if (lane_id() == 0) s_tot[wave_id()] = tot;
  __syncthreads();
#if !(AG2_EXP_NABL & 4)
  if (threadIdx.x == 0) {
    const int t4 = (s_tot[0] + s_tot[1]) + (s_tot[2] + s_tot[3]);
    if (t4) atomicAdd(&st->sum_k1, (unsigned long long)t4);
  }
#endif
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
