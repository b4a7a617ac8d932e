// Host-side guidance for SSLCCT's G-Cutout decoder (pixelssl/ssl_algorithm/ssl_cct.py:615-656).  The reference pulls
// the predicted foreground mask to the host and calls cv2.findContours(RETR_EXTERNAL, CHAIN_APPROX_SIMPLE); OpenCV is a
// third-party dependency, so this is a from-scratch restatement of the published behaviour: outer borders of the
// 8-connected foreground components that are not enclosed by another component, kept when their border polygon (the
// traced border with straight runs collapsed to their end points) has more than `min_vertices` vertices; the result is
// the bounding box of each kept contour in OpenCV's list order -- newest-found first, i.e. REVERSE raster order of the
// contours' first pixels (each finished contour is linked in front of its siblings; the order decides which
// random.randint draw of ssl_cct.py:637-638 lands on which box).  Runs on the CPU like the reference's (the mask is
// 513x513 bytes per sample); parity with OpenCV itself is unpinned (see oracle/cct_oracle.py for the three rules restated
// and tests/test_cct.py for the hand-derived known answers).
#include <vector>
#include <cstdint>
#include <cstring>

#include "common.h"

namespace {

const int NB8[8][2] = {{0, 1}, {1, 1}, {1, 0}, {1, -1}, {0, -1}, {-1, -1}, {-1, 0}, {-1, 1}};   // clockwise from east

// Byte maps hold 0 / 1 per pixel; the run searches below test eight pixels per step.  COMP = false: a pixel is OPEN when it is neither
// foreground nor marked (the outer-background fill); COMP = true: foreground and not marked (the component fill).
constexpr uint64_t ONES8 = 0x0101010101010101ull;
inline uint64_t ld8(const uint8_t* p) { uint64_t v; std::memcpy(&v, p, 8); return v; }
template <bool COMP> inline uint64_t open8(const uint8_t* f, const uint8_t* k) { return COMP ? (ld8(f) & ~ld8(k) & ONES8) : ((ld8(f) | ld8(k)) ^ ONES8) & ONES8; }
template <bool COMP> inline bool open1(const uint8_t* f, const uint8_t* k) { return COMP ? (*f && !*k) : (!*f && !*k); }
// first x in [x, xe] that is open (closed when WANT_OPEN = false), or xe + 1
template <bool COMP, bool WANT_OPEN> inline int next_x(const uint8_t* f, const uint8_t* k, int x, int xe) {
  while (x + 7 <= xe && open8<COMP>(f + x, k + x) == (WANT_OPEN ? 0 : ONES8)) x += 8;
  while (x <= xe && open1<COMP>(f + x, k + x) != WANT_OPEN) ++x;
  return x;
}
// last x in [xb, x] walking left from x while open; returns the leftmost open x of the run that contains x (x itself is open)
template <bool COMP> inline int run_left(const uint8_t* f, const uint8_t* k, int x, int xb) {
  while (x - 8 >= xb && open8<COMP>(f + x - 8, k + x - 8) == ONES8) x -= 8;
  while (x - 1 >= xb && open1<COMP>(f + x - 1, k + x - 1)) --x;
  return x;
}

struct Padded {
  int H, W, P;                       // padded pitch
  std::vector<uint8_t> fg;
  uint8_t at(int y, int x) const { return fg[(size_t)y * P + x]; }
};

int next_from(const Padded& m, int y, int x, int start) {
  for (int k = 0; k < 8; ++k) {
    const int d = (start + k) & 7;
    if (m.at(y + NB8[d][0], x + NB8[d][1])) return d;
  }
  return -1;
}

// Moore border tracing from the raster-first pixel of a component; returns the number of direction changes
long traced_vertices(const Padded& m, int y0, int x0) {
  int d = next_from(m, y0, x0, 5);
  if (d < 0) return 1;
  const int first = d;
  int y = y0, x = x0, prev = -1, first_dir = d;
  long changes = 0, steps = 0;
  const long limit = 8L * m.P * (m.H + 2);
  while (true) {
    if (prev >= 0 && d != prev) ++changes;
    prev = d;
    y += NB8[d][0]; x += NB8[d][1];
    const int nd = next_from(m, y, x, (d + 5) & 7);
    if (y == y0 && x == x0 && nd == first) break;
    d = nd;
    if (++steps > limit) return -1;
  }
  if (prev != first_dir) ++changes;         // closing the chain: last direction vs first
  return changes;
}

}  // namespace

extern "C" int pxl_external_contour_boxes_host(const uint8_t* mask, int H, int W, int min_vertices, int* boxes,
                                               int max_boxes, int* nboxes) {
  PXL_REQUIRE(mask && boxes && nboxes && H > 0 && W > 0 && max_boxes >= 0, "external_contour_boxes_host: bad argument");
  PXL_REQUIRE(H < 65000 && W < 65000, "external_contour_boxes_host: mask too large");
  Padded m;
  m.H = H; m.W = W; m.P = W + 2;
  const int P = m.P;
  m.fg.assign((size_t)(H + 2) * P + 8, 0);
  for (int y = 0; y < H; ++y) {
    const uint8_t* src = mask + (size_t)y * W;
    uint8_t* dst = m.fg.data() + (size_t)(y + 1) * P + 1;
    for (int x = 0; x < W; ++x) dst[x] = src[x] ? 1 : 0;
  }
  // Both flood fills work on horizontal RUNS (the search sat on the training step's critical path, 2.1 ms per 513 x 513 mask with one
  // stack entry and two integer divisions per pixel): a popped seed is extended to its whole unmarked run, which is marked at once, and
  // the rows above and below are scanned -- eight pixels per step -- for the runs that touch it (one entry per run).  Same sets as a
  // pixel-wise fill.
  // background reachable from the frame (4-connectivity)
  const size_t cells = (size_t)(H + 2) * P;
  std::vector<uint8_t> outer(cells + 8, 0);            // (+ 8: the eight-byte loads may start at the last pixel)
  std::vector<uint32_t> stack;
  stack.reserve(4096);
  const uint8_t* fg = m.fg.data();
  {
    uint8_t* mk = outer.data();
    stack.push_back(0);
    while (!stack.empty()) {
      const uint32_t e = stack.back();
      stack.pop_back();
      const int y = (int)(e >> 16), x = (int)(e & 0xffffu);
      const size_t row = (size_t)y * P;
      if (fg[row + x] || mk[row + x]) continue;
      const int xl = run_left<false>(fg + row, mk + row, x, 0);
      const int xr = next_x<false, false>(fg + row, mk + row, x, W + 1) - 1;
      std::memset(mk + row + xl, 1, (size_t)(xr - xl + 1));
      for (int dy = -1; dy <= 1; dy += 2) {
        const int yy = y + dy;
        if (yy < 0 || yy >= H + 2) continue;
        const size_t r2 = (size_t)yy * P;
        for (int xx = next_x<false, true>(fg + r2, mk + r2, xl, xr); xx <= xr;) {
          stack.push_back((uint32_t)yy << 16 | (uint32_t)xx);
          xx = next_x<false, false>(fg + r2, mk + r2, xx, xr);
          xx = next_x<false, true>(fg + r2, mk + r2, xx, xr);
        }
      }
    }
  }
  std::vector<uint8_t> seen(cells + 8, 0);
  uint8_t* sn = seen.data();
  int n = 0;
  for (int y = 1; y <= H; ++y)
    for (int x = next_x<true, true>(fg + (size_t)y * P, sn + (size_t)y * P, 1, W); x <= W;
         x = next_x<true, true>(fg + (size_t)y * P, sn + (size_t)y * P, x + 1, W)) {
      const size_t p0 = (size_t)y * P + x;
      int minx = x, maxx = x, miny = y, maxy = y;
      stack.clear();
      stack.push_back((uint32_t)y << 16 | (uint32_t)x);
      while (!stack.empty()) {                       // 8-connected component of (y, x), run by run
        const uint32_t e = stack.back();
        stack.pop_back();
        const int cy = (int)(e >> 16), cx = (int)(e & 0xffffu);
        const size_t row = (size_t)cy * P;
        if (!fg[row + cx] || sn[row + cx]) continue;
        const int xl = run_left<true>(fg + row, sn + row, cx, 1);       // (columns 0 and W + 1 of the padded map are background)
        const int xr = next_x<true, false>(fg + row, sn + row, cx, W) - 1;
        std::memset(sn + row + xl, 1, (size_t)(xr - xl + 1));
        if (xl < minx) minx = xl;
        if (xr > maxx) maxx = xr;
        if (cy < miny) miny = cy;
        if (cy > maxy) maxy = cy;
        for (int dy = -1; dy <= 1; dy += 2) {         // rows 0 and H + 1 are background: scanning them finds nothing
          const size_t r2 = (size_t)(cy + dy) * P;
          const int lo = xl - 1, hi = xr + 1;         // diagonal neighbours count
          for (int xx = next_x<true, true>(fg + r2, sn + r2, lo, hi); xx <= hi;) {
            stack.push_back((uint32_t)(cy + dy) << 16 | (uint32_t)xx);
            xx = next_x<true, false>(fg + r2, sn + r2, xx, hi);
            xx = next_x<true, true>(fg + r2, sn + r2, xx, hi);
          }
        }
      }
      if (!outer[p0 - 1]) continue;                       // enclosed by another component's hole: not external
      const long nv = traced_vertices(m, y, x);
      if (nv < 0) return pxl_set_error(PXL_ERR_UNSUPPORTED, "external_contour_boxes_host: border tracing did not terminate");
      if (nv <= min_vertices) continue;
      if (n < max_boxes) {
        boxes[4 * n + 0] = minx - 1; boxes[4 * n + 1] = maxx - 1;
        boxes[4 * n + 2] = miny - 1; boxes[4 * n + 3] = maxy - 1;
      }
      ++n;
    }
  // raster order of discovery -> list order (newest first); only the boxes that fitted are stored
  const int stored = n < max_boxes ? n : max_boxes;
  for (int i = 0, j = stored - 1; i < j; ++i, --j)
    for (int k = 0; k < 4; ++k) { const int t = boxes[4 * i + k]; boxes[4 * i + k] = boxes[4 * j + k]; boxes[4 * j + k] = t; }
  *nboxes = n;
  return PXL_OK;
}
