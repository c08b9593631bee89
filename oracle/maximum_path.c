/* Oracle (TEST INFRASTRUCTURE ONLY): plain C restatement of
 * monotonic_align/core.pyx:7-33 (maximum_path_each) and :38-42 (maximum_path_c).
 * Same loops, same float32 arithmetic, `value` mutated in place.  The two
 * out-of-bounds reads the Cython code performs at y == 0 (value[-1, ...] with
 * wraparound disabled) only ever influence `index` after the last path write,
 * so they are guarded here instead of reproduced. */
#include <stdint.h>

static void maximum_path_each(int32_t* path, float* value, int t_y, int t_x, int t_s, float max_neg_val) {
  int x, y, index = t_x - 1;
  float v_prev, v_cur;
  for (y = 0; y < t_y; y++) {
    int lo = t_x + y - t_y; if (lo < 0) lo = 0;
    int hi = y + 1; if (hi > t_x) hi = t_x;
    for (x = lo; x < hi; x++) {
      if (x == y) v_cur = max_neg_val; else v_cur = value[(y - 1) * t_s + x];
      if (x == 0) { if (y == 0) v_prev = 0.f; else v_prev = max_neg_val; }
      else v_prev = value[(y - 1) * t_s + x - 1];
      value[y * t_s + x] += (v_prev > v_cur ? v_prev : v_cur);
    }
  }
  for (y = t_y - 1; y >= 0; y--) {
    if (index >= 0 && index < t_s) path[y * t_s + index] = 1;
    if (index != 0 && y > 0 &&
        (index == y || value[(y - 1) * t_s + index] < value[(y - 1) * t_s + index - 1]))
      index = index - 1;
  }
}

void oracle_maximum_path_c(int32_t* paths, float* values, const int32_t* t_ys, const int32_t* t_xs,
                           int b, int t_t, int t_s) {
  for (int i = 0; i < b; i++)
    maximum_path_each(paths + (long)i * t_t * t_s, values + (long)i * t_t * t_s, t_ys[i], t_xs[i], t_s, -1e9f);
}
