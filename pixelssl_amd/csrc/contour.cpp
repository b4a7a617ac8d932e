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
  Padded m;
  m.H = H; m.W = W; m.P = W + 2;
  m.fg.assign((size_t)(H + 2) * m.P, 0);
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) m.fg[(size_t)(y + 1) * m.P + x + 1] = mask[(size_t)y * W + x] ? 1 : 0;
  // background reachable from the frame (4-connectivity)
  std::vector<uint8_t> outer((size_t)(H + 2) * m.P, 0);
  std::vector<int> stack;
  stack.push_back(0);
  outer[0] = 1;
  const int D4[4][2] = {{0, 1}, {1, 0}, {0, -1}, {-1, 0}};
  while (!stack.empty()) {
    const int p = stack.back();
    stack.pop_back();
    const int y = p / m.P, x = p % m.P;
    for (auto& d : D4) {
      const int yy = y + d[0], xx = x + d[1];
      if (yy < 0 || yy >= H + 2 || xx < 0 || xx >= W + 2) continue;
      const size_t q = (size_t)yy * m.P + xx;
      if (m.fg[q] || outer[q]) continue;
      outer[q] = 1;
      stack.push_back((int)q);
    }
  }
  std::vector<uint8_t> seen((size_t)(H + 2) * m.P, 0);
  int n = 0;
  for (int y = 1; y <= H; ++y)
    for (int x = 1; x <= W; ++x) {
      const size_t p0 = (size_t)y * m.P + x;
      if (!m.fg[p0] || seen[p0]) continue;
      int minx = x, maxx = x, miny = y, maxy = y;
      seen[p0] = 1;
      stack.clear();
      stack.push_back((int)p0);
      while (!stack.empty()) {
        const int p = stack.back();
        stack.pop_back();
        const int cy = p / m.P, cx = p % m.P;
        if (cx < minx) minx = cx;
        if (cx > maxx) maxx = cx;
        if (cy < miny) miny = cy;
        if (cy > maxy) maxy = cy;
        for (auto& d : NB8) {
          const size_t q = (size_t)(cy + d[0]) * m.P + cx + d[1];
          if (m.fg[q] && !seen[q]) { seen[q] = 1; stack.push_back((int)q); }
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
