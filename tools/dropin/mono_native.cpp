// include/viso_mono.h in use: VisualOdometryMono::process(matches) on a list of flow matches read
// from a file, result + Tr_delta + inlier count written back (compared with the Python mirror by
// tests/test_gpu_parity.py).    mono_native matches.bin out.bin f cu cv height pitch iters
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "viso_mono.h"

int main(int argc, char **argv) {
  if (argc < 9) return 2;
  FILE *f = fopen(argv[1], "rb");
  if (!f) return 3;
  int32_t n = 0;
  if (fread(&n, 4, 1, f) != 1) return 3;
  std::vector<Matcher::p_match> m((size_t)n);
  if (n && fread(&m[0], sizeof(Matcher::p_match), (size_t)n, f) != (size_t)n) return 3;
  fclose(f);
  VisualOdometryMono::parameters p;
  p.calib.f = atof(argv[3]);
  p.calib.cu = atof(argv[4]);
  p.calib.cv = atof(argv[5]);
  p.height = atof(argv[6]);
  p.pitch = atof(argv[7]);
  p.ransac_iters = atoi(argv[8]);
  VisualOdometryMono vo(p);
  VisualOdometry &base = vo;
  const bool ok = base.process(m);
  Matrix T = vo.getMotion();
  double rec[18];
  rec[0] = ok ? 1 : 0;
  rec[1] = (double)vo.getNumberOfInliers();
  for (int a = 0; a < 4; a++)
    for (int b = 0; b < 4; b++) rec[2 + a * 4 + b] = T.val[a][b];
  FILE *o = fopen(argv[2], "wb");
  fwrite(rec, sizeof(double), 18, o);
  fclose(o);
  return 0;
}
