// tiny C-ABI driver for debugging on the GPU box: push two synthetic pairs, quad match
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "visomatch.h"
int main(int argc, char **argv) {
  int w = argc > 1 ? atoi(argv[1]) : 320, h = argc > 2 ? atoi(argv[2]) : 128;
  vsm_params p;
  vsm_default_params(&p);
  vsm_handle *m = vsm_create(&p);
  if (!m) return 2;
  std::vector<uint8_t> c((size_t)(w + 64) * h);
  unsigned s = 12345;
  for (auto &v : c) { s = s * 1664525u + 1013904223u; v = (uint8_t)(s >> 24); }
  // crude blur for texture
  for (size_t i = 4; i < c.size(); i++) c[i] = (uint8_t)((c[i] + c[i - 1] + c[i - 2] + c[i - 3]) / 4);
  for (int f = 0; f < 3; f++) {
    int rc = vsm_push_back(m, c.data() + 3 * f + 10, c.data() + 3 * f, w, h, w + 64, 0);
    printf("push %d rc=%d feats:", f, rc);
    for (int k = 0; k < 8; k++) printf(" %d", vsm_num_features(m, k));
    printf("\n");
    rc = vsm_match(m, 2, nullptr);
    printf("match rc=%d n=%d stages %d %d %d %d %d\n", rc, vsm_num_matches(m), vsm_stage_size(m, 0), vsm_stage_size(m, 1),
           vsm_stage_size(m, 2), vsm_stage_size(m, 3), vsm_stage_size(m, 4));
  }
  vsm_destroy(m);
  return 0;
}
