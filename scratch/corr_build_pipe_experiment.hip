// scratch (not compiled into the library): the persistent, wave-specialised form of the fused volume build that was
// measured against corr_build_fused_kernel<2> on 2026-09-27 and dropped.  It fits into csrc/corr_build_fused.hip right
// after the classic kernel (same helpers: FT_ROWS, half8, float16v, FusedLevels, pool4, lds_barrier).
// Result (32 edges, 64x64, C = 128): 20.5-21.5 us/edge against 19.8 for the classic kernel with the same store loops;
// 55x55: 17.7 vs 17.1; 48x64: 11.7-12.2 vs 11.8-12.0.  Counters (scratch/pipe_pmc.sh): VALU busy 28 %, LDS busy 17 %,
// waves waiting 63 % of their cycles: the two roles mostly wait for each other at the three barriers (the compute
// waves' MFMA phase, 8-9 k cycles per tile with 8 waves on the CU, and the store waves' serial tail are both on the
// critical path), so the overlap of stores with multiplications that the design is about never materialises.
// ---------------------------------------------------------------------------------------------------------------------
// Pipelined form for the maps every configuration but KITTI uses (w2 <= 64, C = 128): persistent workgroups, one per CU.
//
// What the classic kernel above loses (scratch/cu_rates.hip, header of this file): all workgroups run in phase, so the
// chip alternates between a phase in which nobody stores and one in which every CU stores and HBM's write side
// (5.4-5.6 TB/s for the whole chip = 9 B/clk/CU, where one CU alone sustains 28 B/clk) is the only thing working.
// Here the stores of tile i - 1 leave WHILE tile i is multiplied:
//   * waves 0..7 (compute) : source operand of tile i -> LDS, 32 MFMAs each (target fragments through a 4-deep register
//     ring, the first half of the next tile's requested before the current tile is written), f16 tile -> T[i & 1];
//   * waves 8..15 (store)  : everything after the tile write of the classic kernel, for tile i - 1 out of T[(i-1) & 1]:
//     level-0 stores, 8 x 8 pooling, pooled levels staged over the dead tile, their stores.
//   The compute waves never issue a store and the store waves never wait for a load: on gfx9 a wave's loads and stores
//   complete in order through one counter, which is why interleaving both in the same wave only adds their times up.
//   Three LDS-only barriers per iteration, placed so that neither role waits for the other's long phase:
//     x: source operand in LDS            compute: MFMAs, tile write        store: level-0 stores + pooling of the old tile
//     a: new tile written, old tile read  compute: -                        store: pooled levels over the old tile
//     b: pooled levels staged             compute: next source operand      store: pooled-level stores
// LDS: 2 x 66 KB tiles + 16 KB source operand = 148 KB.  Tiles are dealt round-robin (tile = blockIdx.x + i * gridDim.x,
// strip fastest), so the chip works on half an edge at a time and both feature maps stay in L2.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int PIPE_THREADS = 1024;
constexpr int PIPE_C = 128;
constexpr int PIPE_W2P = 64;
constexpr int PIPE_RP = PIPE_W2P + 4;                // tile row: w2 columns, then columns 0..3 once more (see the store waves)
constexpr int PIPE_PITCH = FT_ROWS * PIPE_RP + 4;
constexpr size_t PIPE_LDS_BYTES = sizeof(_Float16) * ((size_t)2 * 64 * PIPE_PITCH + (size_t)64 * PIPE_C);

__global__ __launch_bounds__(PIPE_THREADS) void corr_build_pipe_kernel(const _Float16 *__restrict__ A,
                                                                       const _Float16 *__restrict__ Bm, FusedLevels L,
                                                                       int h1, int w1, int h2, int w2, int HW1p,
                                                                       float inv_w1, int nstrips, int nrt, int ntiles
#ifdef FB_PROF
                                                                       , unsigned long long *prof
#endif
) {
#ifdef FB_PROF
  unsigned long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, plast = __builtin_amdgcn_s_memtime();
#define PP(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); pacc[k] += t_ - plast; plast = t_; } while (0)
#define PP_FLUSH(role) do { if ((threadIdx.x & 63) == 0 && ((threadIdx.x >> 6) & 7) == 0) for (int k_ = 0; k_ < 8; k_++) prof[((size_t)blockIdx.x * 2 + (role)) * 8 + k_] = pacc[k_]; } while (0)
#else
#define PP(k) (void)0
#define PP_FLUSH(role) (void)0
#endif
  constexpr int C = PIPE_C, W2P = PIPE_W2P, RP = PIPE_RP, PITCH = PIPE_PITCH, KS = C / 16, NT = 2;
  extern __shared__ __attribute__((aligned(16))) _Float16 smem[];
  _Float16 *Abuf = smem + 2 * 64 * PITCH;  // [KS][64 pixels][16] halves
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int HW1 = h1 * w1, HW2 = h2 * w2;
  const int nt = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;  // tiles of this workgroup (>= 1)
  auto decode = [&](int i, int &p0, int &ty0, int &e) {
    const int idx = (int)blockIdx.x + i * (int)gridDim.x;
    const int strip = idx % nstrips, rest = idx / nstrips;
    p0 = strip * 64;
    ty0 = (rest % nrt) * FT_ROWS;
    e = rest / nrt;
  };

  if (wave < 8) {
    // ================================================ compute waves ================================================
    const int l31 = lane & 31, kh = (lane >> 5) * 8;
    half8 apre[2];        // this thread's two 16-byte pieces of the next source operand
    half8 ring[4][NT];    // target fragments of four k-steps
    const _Float16 *bp[NT];
    auto request_a = [&](int p0, int e) {
      const _Float16 *Ae = A + (size_t)e * HW1 * C;
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const int idx = tid + 512 * u, kbk = idx >> 7, r = idx & 127, px = r >> 1, hf = r & 1;
        apre[u] = *reinterpret_cast<const half8 *>(Ae + ((size_t)kbk * HW1 + min(p0 + px, HW1 - 1)) * 16 + hf * 8);
      }
    };
    auto set_b = [&](int ty0, int e) {
      const int ty = min(ty0 + wave, h2 - 1);  // rows past the map are computed on a valid row and never stored
#pragma unroll
      for (int t = 0; t < NT; t++)
        bp[t] = Bm + (size_t)e * HW2 * C + (size_t)min(ty * w2 + t * 32 + l31, HW2 - 1) * 8 + (size_t)(lane >> 5) * HW2 * 8;
    };
    auto request_b = [&](int slot, int ks) {
#pragma unroll
      for (int t = 0; t < NT; t++) ring[slot][t] = *reinterpret_cast<const half8 *>(bp[t] + (size_t)ks * 16 * HW2);
    };
    {
      int p0, ty0, e;
      decode(0, p0, ty0, e);
      request_a(p0, e);
      set_b(ty0, e);
#pragma unroll
      for (int ks = 0; ks < 4; ks++) request_b(ks, ks);
    }
    for (int i = 0; i <= nt; i++) {
      if (i < nt) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
          const int idx = tid + 512 * u, kbk = idx >> 7, r = idx & 127, px = r >> 1, hf = r & 1;
          *reinterpret_cast<half8 *>(Abuf + (kbk * 64 + px) * 16 + hf * 8) = apre[u];
        }
      }
      PP(0);
      lds_barrier();  // x
      PP(1);
      if (i < nt) {
        float16v acc[2][NT];
#pragma unroll
        for (int ii = 0; ii < 2; ii++)
#pragma unroll
          for (int j = 0; j < NT; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[ii][j][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
          half8 a[2];
#pragma unroll
          for (int t = 0; t < 2; t++) a[t] = *reinterpret_cast<const half8 *>(Abuf + (ks * 64 + t * 32 + l31) * 16 + kh);
#pragma unroll
          for (int ii = 0; ii < 2; ii++)
#pragma unroll
            for (int j = 0; j < NT; j++)
              acc[ii][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[ks & 3][j], a[ii], acc[ii][j], 0, 0, 0);  // targets x sources
          if (ks + 4 < KS) request_b(ks & 3, ks + 4);
        }
        PP(2);
        if (i + 1 < nt) {  // the next tile's operands travel while this one is rounded and written
          int p0, ty0, e;
          decode(i + 1, p0, ty0, e);
          request_a(p0, e);
          set_b(ty0, e);
#pragma unroll
          for (int ks = 0; ks < 4; ks++) request_b(ks, ks);
        }
        _Float16 *T = smem + (i & 1) * 64 * PITCH;
#pragma unroll
        for (int ii = 0; ii < 2; ii++)
#pragma unroll
          for (int j = 0; j < NT; j++)
#pragma unroll
            for (int rq = 0; rq < 4; rq++) {
              const int src = ii * 32 + l31;
              const int tx = j * 32 + 8 * rq + 4 * (lane >> 5);
              half4 v;
#pragma unroll
              for (int q = 0; q < 4; q++) v[q] = (_Float16)acc[ii][j][4 * rq + q];
              *reinterpret_cast<half4 *>(T + src * PITCH + wave * RP + tx) = v;
            }
        // columns 0..3 once more behind column w2 - 1 (after the row's own writes: for w2 < 64 those put unused targets
        // there): a store lane's four diagonal reads then never wrap inside a quad
        if (lane < 32) {
#pragma unroll
          for (int ii = 0; ii < 2; ii++) {
            _Float16 *wr = T + (ii * 32 + l31) * PITCH + wave * RP + w2;
#pragma unroll
            for (int q = 0; q < 4; q++) wr[q] = (_Float16)acc[ii][0][q];
          }
        }
      }
      PP(3);
      lds_barrier();  // a
      PP(4);
      lds_barrier();  // b
      PP(5);
    }
    PP_FLUSH(0);
    return;
  }

  // ================================================== store waves ==================================================
  const int sw = wave - 8, stid = tid - 512;  // 8 waves, 512 threads
  constexpr unsigned OOR = 0x80000000u;
  const unsigned plane_bytes = 2u * (unsigned)HW1p;
  auto pixel_xy = [&](int pix, int &x, int &y) {
    const int pc = min(pix, HW1 - 1);
    y = (int)(((float)pc + 0.5f) * inv_w1);
    x = pc - y * w1;
    if (x < 0) { y--; x += w1; }
    if (x >= w1) { y++; x -= w1; }
  };
  const int q4 = (lane & 15) * 4, g = lane >> 4;
  constexpr int NBLK = W2P / 8;            // 8-column blocks per pixel
  constexpr int PER = 64 * NBLK / 512;     // blocks per store thread
  _Float16 q1[PER][4][4], q2[PER][2][2], q3[PER];
  for (int i = 0; i <= nt; i++) {
    int p0 = 0, ty0 = 0, e = 0;
    if (i >= 1) decode(i - 1, p0, ty0, e);
    _Float16 *T = smem + ((i + 1) & 1) * 64 * PITCH;  // the tile of iteration i - 1
    const int p = p0 + lane;
    const bool active = p < HW1;
    int x1, y1, qx[4], qy[4];
    pixel_xy(p, x1, y1);
#pragma unroll
    for (int u = 0; u < 4; u++) pixel_xy(p0 + q4 + u, qx[u], qy[u]);
    const bool quad_regular = (p0 + q4 + 3 < HW1) && (qy[0] == qy[3]);
    auto level_rsrc = [&](int lvl) {
      const size_t elems = (size_t)(h2 >> lvl) * (w2 >> lvl) * HW1p;
      return __builtin_amdgcn_make_buffer_rsrc((void *)(L.vs[lvl] + (size_t)e * elems), 0, (int)(2 * elems), 0x00020000);
    };
    auto store_quad = [&](const __amdgpu_buffer_rsrc_t &rl, int lvl, int tyg, int dx, int h2l, int w2l, _Float16 v0, _Float16 v1,
                          _Float16 v2, _Float16 v3, bool on) {
      const _Float16 vv[4] = {v0, v1, v2, v3};
      if (quad_regular) {
        int dy = tyg - (qy[0] >> lvl);
        dy += (dy < 0) ? h2l : 0;
        const unsigned voff = on ? ((unsigned)dy * (unsigned)w2l + (unsigned)dx) * plane_bytes + 2u * (unsigned)(p0 + q4) : OOR;
        typedef unsigned u2v __attribute__((ext_vector_type(2)));
        u2v d;
        d.x = (unsigned)__builtin_bit_cast(unsigned short, v0) | ((unsigned)__builtin_bit_cast(unsigned short, v1) << 16);
        d.y = (unsigned)__builtin_bit_cast(unsigned short, v2) | ((unsigned)__builtin_bit_cast(unsigned short, v3) << 16);
        __builtin_amdgcn_raw_buffer_store_b64(d, rl, voff, 0, 0);
      } else {
#pragma unroll
        for (int u = 0; u < 4; u++) {
          int dy = tyg - (qy[u] >> lvl);
          dy += (dy < 0) ? h2l : 0;
          const bool ok = on && (p0 + q4 + u < HW1);
          const unsigned voff = ok ? ((unsigned)dy * (unsigned)w2l + (unsigned)dx) * plane_bytes + 2u * (unsigned)(p0 + q4 + u) : OOR;
          __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, vv[u]), rl, voff, 0, 0);
        }
      }
    };
    PP(0);
    lds_barrier();  // x
    PP(1);
    if (i >= 1) {
      // ---- level 0: store wave w owns target row ty0 + w.  A lane's quad of pixels reads the tile along a diagonal
      // (pixel + 1, column + 1); with columns 0..3 replicated behind the row only the first column wraps, once per
      // iteration, and the store offset just advances by four planes.
      const int ty = ty0 + sw;
      if (ty < h2) {  // (wave-uniform)
        const __amdgpu_buffer_rsrc_t r0 = level_rsrc(0);
        if (quad_regular) {
          int t = qx[0] + g;
          t -= (t >= w2) ? w2 : 0;
          int dy = ty - qy[0];
          dy += (dy < 0) ? h2 : 0;
          unsigned voff = ((unsigned)dy * (unsigned)w2 + (unsigned)g) * plane_bytes + 2u * (unsigned)(p0 + q4);
          const _Float16 *lb = T + q4 * PITCH + sw * RP;
          // four lines of offsets per batch: the 16 LDS reads of a batch are in flight together (one LDS round trip per
          // line otherwise: the store waves' critical path)
          for (int dx0 = 0; dx0 < w2; dx0 += 16) {
            unsigned short a[4][4];
            int tt = t;
#pragma unroll
            for (int b = 0; b < 4; b++) {
              const _Float16 *pp = lb + tt;
#pragma unroll
              for (int u = 0; u < 4; u++) a[b][u] = __builtin_bit_cast(unsigned short, pp[u * (PITCH + 1)]);
              tt += 4;
              tt -= (tt >= w2) ? w2 : 0;
            }
            t = tt;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int b = 0; b < 4; b++) {
              typedef unsigned u2v __attribute__((ext_vector_type(2)));
              u2v d;
              d.x = (unsigned)a[b][0] | ((unsigned)a[b][1] << 16);
              d.y = (unsigned)a[b][2] | ((unsigned)a[b][3] << 16);
#ifdef PIPE_ABLATE_L0_OOR
              __builtin_amdgcn_raw_buffer_store_b64(d, r0, OOR, 0, 0);
#else
              __builtin_amdgcn_raw_buffer_store_b64(d, r0, (dx0 + 4 * b + g < w2) ? voff : OOR, 0, 0);
#endif
              voff += 4u * plane_bytes;
            }
          }
        }
        if (!quad_regular) {  // a quad that spans a row end (or the end of the map): four 2-byte stores per line
          const _Float16 *rowp = T + q4 * PITCH + sw * RP;
          for (int dx0 = 0; dx0 < w2; dx0 += 4) {
            const int dx = dx0 + g;
            _Float16 v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
              int tx = qx[u] + dx;
              tx -= (tx >= w2) ? w2 : 0;
              tx = min(tx, w2 - 1);  // (dx beyond the map in the last group of four: read something valid, store nothing)
              v[u] = rowp[u * PITCH + tx];
            }
            store_quad(r0, 0, ty, dx, h2, w2, v[0], v[1], v[2], v[3], dx < w2);
          }
        }
      }
      PP(2);
      // ---- 8 x 8 blocks -> 4 x 4, 2 x 2, 1 (each from the ROUNDED level below), in registers
#pragma unroll
      for (int u = 0; u < PER; u++) {
        const int blk = stid + 512 * u, src = blk / NBLK, cb = blk - src * NBLK;
        const _Float16 *tb = T + src * PITCH + 8 * cb;
        _Float16 t8[8][8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
          const half4 lo = *reinterpret_cast<const half4 *>(tb + r * RP), hi = *reinterpret_cast<const half4 *>(tb + r * RP + 4);
#pragma unroll
          for (int c = 0; c < 4; c++) t8[r][c] = lo[c], t8[r][4 + c] = hi[c];
        }
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
          for (int c = 0; c < 4; c++) q1[u][r][c] = pool4(t8[2 * r][2 * c], t8[2 * r][2 * c + 1], t8[2 * r + 1][2 * c], t8[2 * r + 1][2 * c + 1]);
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
          for (int c = 0; c < 2; c++)
            q2[u][r][c] = pool4(q1[u][2 * r][2 * c], q1[u][2 * r][2 * c + 1], q1[u][2 * r + 1][2 * c], q1[u][2 * r + 1][2 * c + 1]);
        q3[u] = pool4(q2[u][0][0], q2[u][0][1], q2[u][1][0], q2[u][1][1]);
      }
    }
    PP(3);
    lds_barrier();  // a: every read of the old tile is done
    PP(4);
    constexpr int RP1 = W2P / 2 + 4;                 // level-1 row: w2 >> 1 columns, then columns 0..3 once more
    _Float16 *P1 = T;                                // [64][4][RP1]
    _Float16 *P2 = P1 + 64 * 4 * RP1;                // [64][2][W2P / 4]
    _Float16 *P3 = P2 + 64 * 2 * (W2P / 4);          // [64][W2P / 8]
    if (i >= 1) {
#pragma unroll
      for (int u = 0; u < PER; u++) {
        const int blk = stid + 512 * u, src = blk / NBLK, cb = blk - src * NBLK;
#pragma unroll
        for (int r = 0; r < 4; r++) {
          half4 v;
#pragma unroll
          for (int c = 0; c < 4; c++) v[c] = q1[u][r][c];
          _Float16 *row1 = P1 + (src * 4 + r) * RP1;
          if (4 * cb + 4 <= (w2 >> 1)) {
            *reinterpret_cast<half4 *>(row1 + 4 * cb) = v;
          } else {  // the block that holds column (w2 >> 1) - 1: the columns behind it belong to the copy of block 0
#pragma unroll
            for (int c = 0; c < 4; c++)
              if (4 * cb + c < (w2 >> 1)) row1[4 * cb + c] = q1[u][r][c];
          }
          if (cb == 0) {
#pragma unroll
            for (int c = 0; c < 4; c++) row1[(w2 >> 1) + c] = q1[u][r][c];
          }
        }
#pragma unroll
        for (int r = 0; r < 2; r++) {
          half2v v;
          v.x = q2[u][r][0], v.y = q2[u][r][1];
          *reinterpret_cast<half2v *>(P2 + (src * 2 + r) * (W2P / 4) + 2 * cb) = v;
        }
        P3[src * (W2P / 8) + cb] = q3[u];
      }
    }
    PP(5);
    lds_barrier();  // b
    PP(6);
    if (i >= 1) {
      {  // level 1: 4 rows x (w2 >> 1) offsets, four lines per store instruction; wave w takes row w & 3 and every
         // other group of four offsets.  The quad's columns (x >> 1) - (x0 >> 1) are 0, 0|1, 1, 1|2: lane constants
        const int w2l = w2 >> 1, h2l = h2 >> 1;
        const __amdgpu_buffer_rsrc_t rl = level_rsrc(1);
        const int tyl = sw & 3, tyg = (ty0 >> 1) + tyl;
        if (tyg < h2l) {  // floor sizes of avg_pool2d: the last partial row of the level below is dropped
          if (quad_regular) {
            const int xh = qx[0] >> 1;
            const int o1 = (qx[1] >> 1) - xh, o2 = (qx[2] >> 1) - xh, o3 = (qx[3] >> 1) - xh;
            int t = xh + 4 * (sw >> 2) + g;
            t -= (t >= w2l) ? w2l : 0;
            t -= (t >= w2l) ? w2l : 0;  // (maps down to 8 columns: twice)
            int dy = tyg - (qy[0] >> 1);
            dy += (dy < 0) ? h2l : 0;
            unsigned voff = ((unsigned)dy * (unsigned)w2l + (unsigned)(4 * (sw >> 2) + g)) * plane_bytes + 2u * (unsigned)(p0 + q4);
            const _Float16 *lb = P1 + (q4 * 4 + tyl) * RP1;
            for (int dx0 = 4 * (sw >> 2); dx0 < w2l; dx0 += 8) {
              const _Float16 *pp = lb + t;
              const unsigned short a0 = __builtin_bit_cast(unsigned short, pp[0]), a1 = __builtin_bit_cast(unsigned short, pp[4 * RP1 + o1]);
              const unsigned short a2 = __builtin_bit_cast(unsigned short, pp[8 * RP1 + o2]), a3 = __builtin_bit_cast(unsigned short, pp[12 * RP1 + o3]);
              typedef unsigned u2v __attribute__((ext_vector_type(2)));
              u2v d;
              d.x = (unsigned)a0 | ((unsigned)a1 << 16);
              d.y = (unsigned)a2 | ((unsigned)a3 << 16);
              __builtin_amdgcn_raw_buffer_store_b64(d, rl, (dx0 + g < w2l) ? voff : OOR, 0, 0);
              voff += 8u * plane_bytes;
              t += 8;
              t -= (t >= w2l) ? w2l : 0;
              t -= (t >= w2l) ? w2l : 0;
            }
          }
          if (!quad_regular) {
            for (int dx0 = 4 * (sw >> 2); dx0 < w2l; dx0 += 8) {
              const int dx = dx0 + g;
              _Float16 v[4];
#pragma unroll
              for (int u = 0; u < 4; u++) {
                int tx = (qx[u] >> 1) + dx;
                tx -= (tx >= w2l) ? w2l : 0;
                tx = min(tx, w2l - 1);
                v[u] = P1[((q4 + u) * 4 + tyl) * RP1 + tx];
              }
              store_quad(rl, 1, tyg, dx, h2l, w2l, v[0], v[1], v[2], v[3], dx < w2l);
            }
          }
        }
      }
      auto store_level = [&](int lvl, const _Float16 *Pl, int rows, int pitch_cols) {
        const int h2l = h2 >> lvl, w2l = w2 >> lvl;
        const __amdgpu_buffer_rsrc_t rl = level_rsrc(lvl);
        const int x1l = x1 >> lvl, y1l = y1 >> lvl;
        for (int seg = sw; seg < rows * w2l; seg += 8) {  // (wave-uniform)
          const int tyl = seg / w2l, dx = seg - tyl * w2l;
          const int tyg = (ty0 >> lvl) + tyl;
          if (tyg >= h2l) continue;
          int dy = tyg - y1l;
          dy += (dy < 0) ? h2l : 0;
          int tx = x1l + dx;
          tx -= (tx >= w2l) ? w2l : 0;
          const _Float16 v = Pl[(lane * rows + tyl) * pitch_cols + tx];
          const unsigned voff = active ? ((unsigned)dy * (unsigned)w2l + (unsigned)dx) * plane_bytes + 2u * (unsigned)p : OOR;
          __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, v), rl, voff, 0, 0);
        }
      };
      store_level(2, P2, 2, W2P / 4);
      store_level(3, P3, 1, W2P / 8);
    }
  }
  PP(7);
  PP_FLUSH(1);
}

