// Visit statistics of the blend walk under different wave decompositions: a CPU replay of the oracle's per-tile lists
// (driven by tools/sim_tile_order.py; measurement tooling, not part of the product).
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <stdio.h>
#include <omp.h>
// out[0]=quadrant visits (>=1 hit), out[1]=sum over (wave,batch) max over 4 4x4 blocks, out[2]= same for 8x2 rows,
// out[3]=sum over (wave,batch) max over 2 8x4 halves, out[4]=lane hits total, out[5]=sum of 4x4 block visits,
// out[6]=sum over waves (no batching) of max over 4x4 blocks, out[7]=# (tile,batch) rounds,
// out[8]=sum over (tile,batch) of max over 16 blocks (whole-WG lockstep), out[9]=sum 8x4 half visits,
// out[10]=sum over rounds of the busiest quadrant's visits, out[11]=sum over rounds of the mean over the four quadrants,
// out[12]=quadrant visits whose hits lie in one 32-lane half, out[13]=... in one 16-lane row
void replay(int W, int H, const uint32_t* ranges, const uint32_t* plist, const float* m2d, const float* co,
            const uint32_t* ncontrib, int batch, double* out, float* tile_cost) {
  int gx = (W + 15) / 16, gy = (H + 15) / 16;
  double acc[16] = {0};
#pragma omp parallel
  {
    double a[16] = {0};
#pragma omp for schedule(dynamic, 8)
    for (int t = 0; t < gx * gy; ++t) {
      uint32_t r0 = ranges[2 * t], r1 = ranges[2 * t + 1];
      if (r1 <= r0) continue;
      int tx = t % gx, ty = t / gx;
      uint32_t nc[256]; uint32_t maxc = 0;
      for (int p = 0; p < 256; ++p) {
        int px = tx * 16 + (p & 15), py = ty * 16 + (p >> 4);
        nc[p] = (px < W && py < H) ? ncontrib[py * W + px] : 0;
        if (nc[p] > maxc) maxc = nc[p];
      }
      int tot44[4][4] = {{0}}; double tcost = 0;
      for (uint32_t b0 = 0; b0 < maxc; b0 += batch) {
        int c44[4][4] = {{0}}, c82[4][4] = {{0}}, c84[4][2] = {{0}}; int cq[4] = {0,0,0,0};
        a[7] += 1;
        for (uint32_t pos = b0; pos < b0 + batch && pos < maxc; ++pos) {
          uint32_t g = plist[r0 + pos];
          float X = m2d[2 * g], Y = m2d[2 * g + 1], A = co[4 * g], B = co[4 * g + 1], Cc = co[4 * g + 2], O = co[4 * g + 3];
          uint32_t h44 = 0, h82 = 0, h84 = 0, hq = 0; int lanes = 0;
          for (int p = 0; p < 256; ++p) {
            if (pos >= nc[p]) continue;
            int lx = p & 15, ly = p >> 4;
            float dx = X - (float)(tx * 16 + lx), dy = Y - (float)(ty * 16 + ly);
            float pw = -0.5f * (A * dx * dx + Cc * dy * dy) - B * dx * dy;
            if (pw > 0.f) continue;
            float al = fminf(0.99f, O * expf(pw));
            if (al < 1.f / 255.f) continue;
            int q = (ly >> 3) * 2 + (lx >> 3), qx = lx & 7, qy = ly & 7;
            hq |= 1u << q; ++lanes;
            h44 |= 1u << (q * 4 + (qy >> 2) * 2 + (qx >> 2));
            h82 |= 1u << (q * 4 + (qy >> 1));
            h84 |= 1u << (q * 2 + (qy >> 2));
          }
          a[4] += lanes;
          for (int q = 0; q < 4; ++q) {
            if (hq >> q & 1) {
              a[0] += 1; cq[q]++;
              // hits confined to one 32-lane half (pixel rows 0-3 or 4-7 of the quadrant) / to one 16-lane DPP row: the
              // cross-half (cross-row) stage of the wave reduction would have nothing to add for such a visit
              const uint32_t hv = h84 >> (q * 2) & 3u, rv = h82 >> (q * 4) & 15u;
              if (hv == 1u || hv == 2u) a[12] += 1;
              if ((rv & (rv - 1u)) == 0u) a[13] += 1;
            }
            for (int k = 0; k < 4; ++k) { c44[q][k] += h44 >> (q * 4 + k) & 1; c82[q][k] += h82 >> (q * 4 + k) & 1; }
            for (int k = 0; k < 2; ++k) c84[q][k] += h84 >> (q * 2 + k) & 1;
          }
        }
        int m16 = 0, mq = 0; { int cq[4]={0,0,0,0}; (void)cq; }
        for (int q = 0; q < 4; ++q) {
          int m = 0, m2 = 0, m3 = 0;
          for (int k = 0; k < 4; ++k) { if (c44[q][k] > m) m = c44[q][k]; if (c82[q][k] > m2) m2 = c82[q][k]; a[5] += c44[q][k]; tot44[q][k] += c44[q][k]; }
          for (int k = 0; k < 2; ++k) { if (c84[q][k] > m3) m3 = c84[q][k]; a[9] += c84[q][k]; }
          a[1] += m; a[2] += m2; a[3] += m3; if (m > m16) m16 = m;
        }
        a[8] += m16; for (int q = 0; q < 4; ++q) if (cq[q] > mq) mq = cq[q]; tcost += mq + 6;
        a[10] += mq; a[11] += 0.25 * (cq[0] + cq[1] + cq[2] + cq[3]);  // barrier skew of a round: max vs mean over the quadrants
      }
      for (int q = 0; q < 4; ++q) { int m = 0; for (int k = 0; k < 4; ++k) if (tot44[q][k] > m) m = tot44[q][k]; a[6] += m; }
      tile_cost[t] = (float)tcost;
    }
#pragma omp critical
    for (int i = 0; i < 16; ++i) acc[i] += a[i];
  }
  for (int i = 0; i < 16; ++i) out[i] = acc[i];
}

// Histogram of the number of hitting lanes per (quadrant, instance) visit: hist[k] = visits with exactly k of the 64 pixels
// of the quadrant passing the alpha test while still unfinished (k = 1..64).  Same replay as above.
void replay_lane_hist(int W, int H, const uint32_t* ranges, const uint32_t* plist, const float* m2d, const float* co,
                      const uint32_t* ncontrib, double* hist) {
  int gx = (W + 15) / 16, gy = (H + 15) / 16;
  double acc[65] = {0};
#pragma omp parallel
  {
    double a[65] = {0};
#pragma omp for schedule(dynamic, 8)
    for (int t = 0; t < gx * gy; ++t) {
      uint32_t r0 = ranges[2 * t], r1 = ranges[2 * t + 1];
      if (r1 <= r0) continue;
      int tx = t % gx, ty = t / gx;
      uint32_t nc[256]; uint32_t maxc = 0;
      for (int p = 0; p < 256; ++p) {
        int px = tx * 16 + (p & 15), py = ty * 16 + (p >> 4);
        nc[p] = (px < W && py < H) ? ncontrib[py * W + px] : 0;
        if (nc[p] > maxc) maxc = nc[p];
      }
      for (uint32_t pos = 0; pos < maxc; ++pos) {
        uint32_t g = plist[r0 + pos];
        float X = m2d[2 * g], Y = m2d[2 * g + 1], A = co[4 * g], B = co[4 * g + 1], Cc = co[4 * g + 2], O = co[4 * g + 3];
        int lanes[4] = {0, 0, 0, 0};
        for (int p = 0; p < 256; ++p) {
          if (pos >= nc[p]) continue;
          int lx = p & 15, ly = p >> 4;
          float dx = X - (float)(tx * 16 + lx), dy = Y - (float)(ty * 16 + ly);
          float pw = -0.5f * (A * dx * dx + Cc * dy * dy) - B * dx * dy;
          if (pw > 0.f) continue;
          if (fminf(0.99f, O * expf(pw)) < 1.f / 255.f) continue;
          lanes[(ly >> 3) * 2 + (lx >> 3)]++;
        }
        for (int q = 0; q < 4; ++q) if (lanes[q]) a[lanes[q]] += 1;
      }
    }
#pragma omp critical
    for (int i = 0; i < 65; ++i) acc[i] += a[i];
  }
  for (int i = 0; i < 65; ++i) hist[i] = acc[i];
}

// What a ROW-GRANULAR backward walk would cost: every 16-lane row of a quadrant's wave owns a block of
// pixels (shape 0: 4x4, shape 1: 8x2 strips) and walks only the instances that hit ITS block.  Rounds of `round_n` entries of
// the tile's list that have at least one hit (the compact hit list).  out[0] = quadrant visits (today's walk: one step each),
// out[1] = block visits, out[2] = steps of the row walk (per round and wave: the busiest row's visits),
// out[3] = sum over rounds of the busiest WAVE's steps x 4 today, out[4] = the same for the row walk (barrier skew included),
// out[5] = list entries with at least one hit (the compact hit list's length), out[6] = rounds whose quadrant visits exceed
// 2 * round_n (the visit rows a round of the shipped kernel has).
void replay_row_walk(int W, int H, const uint32_t* ranges, const uint32_t* plist, const float* m2d, const float* co,
                     const uint32_t* ncontrib, int shape, int round_n, double* out) {
  int gx = (W + 15) / 16, gy = (H + 15) / 16;
  double acc[7] = {0};
#pragma omp parallel
  {
    double a[7] = {0};
#pragma omp for schedule(dynamic, 8)
    for (int t = 0; t < gx * gy; ++t) {
      uint32_t r0 = ranges[2 * t], r1 = ranges[2 * t + 1];
      if (r1 <= r0) continue;
      int tx = t % gx, ty = t / gx;
      uint32_t nc[256]; uint32_t maxc = 0;
      for (int p = 0; p < 256; ++p) {
        int px = tx * 16 + (p & 15), py = ty * 16 + (p >> 4);
        nc[p] = (px < W && py < H) ? ncontrib[py * W + px] : 0;
        if (nc[p] > maxc) maxc = nc[p];
      }
      int in_round = 0;
      int vis_w[4] = {0, 0, 0, 0}, row_w[4][4] = {{0}};
      // the backward walks back to front; the round structure is the same counted from either end up to the remainder
      for (int64_t pos = (int64_t)maxc - 1; pos >= -1; --pos) {
        int flush = pos < 0;
        if (!flush) {
          uint32_t g = plist[r0 + pos];
          float X = m2d[2 * g], Y = m2d[2 * g + 1], A = co[4 * g], B = co[4 * g + 1], Cc = co[4 * g + 2], O = co[4 * g + 3];
          int blk[4][4] = {{0}};
          int any = 0;
          for (int p = 0; p < 256; ++p) {
            if ((uint32_t)pos >= nc[p]) continue;
            int lx = p & 15, ly = p >> 4;
            float dx = X - (float)(tx * 16 + lx), dy = Y - (float)(ty * 16 + ly);
            float pw = -0.5f * (A * dx * dx + Cc * dy * dy) - B * dx * dy;
            if (pw > 0.f) continue;
            if (fminf(0.99f, O * expf(pw)) < 1.f / 255.f) continue;
            int q = (ly >> 3) * 2 + (lx >> 3), qx = lx & 7, qy = ly & 7;
            int r = shape == 0 ? (qy >> 2) * 2 + (qx >> 2) : (qy >> 1);
            blk[q][r] = 1;
            any = 1;
          }
          if (any) {
            for (int q = 0; q < 4; ++q) {
              int v = blk[q][0] | blk[q][1] | blk[q][2] | blk[q][3];
              vis_w[q] += v;
              for (int r = 0; r < 4; ++r) row_w[q][r] += blk[q][r];
            }
            in_round++;
          }
        }
        if ((in_round == round_n || flush) && in_round > 0) {
          int mx_vis = 0, mx_steps = 0;
          a[5] += in_round;
          if (vis_w[0] + vis_w[1] + vis_w[2] + vis_w[3] > 2 * round_n) a[6] += 1;
          for (int q = 0; q < 4; ++q) {
            int st = 0;
            for (int r = 0; r < 4; ++r) { a[1] += row_w[q][r]; if (row_w[q][r] > st) st = row_w[q][r]; }
            a[0] += vis_w[q];
            a[2] += st;
            if (vis_w[q] > mx_vis) mx_vis = vis_w[q];
            if (st > mx_steps) mx_steps = st;
            vis_w[q] = 0;
            for (int r = 0; r < 4; ++r) row_w[q][r] = 0;
          }
          a[3] += 4.0 * mx_vis;
          a[4] += 4.0 * mx_steps;
          in_round = 0;
        }
      }
    }
#pragma omp critical
    for (int i = 0; i < 7; ++i) acc[i] += a[i];
  }
  for (int i = 0; i < 7; ++i) out[i] = acc[i];
}
