// Drop-in proof: the reference's own VisualOdometryStereo (viso/viso_stereo.cpp, viso/viso.cpp,
// viso/matrix.cpp, compiled unmodified where they lie) running on top of include/matcher.h +
// libvisomatch.so instead of the reference's matcher.cpp / filter.cpp / triangle.cpp.
//
//   vo_dropin frames.raw out.bin f cu cv base
// frames.raw: int32 w,h,n then n x {left,right} x h x w bytes.  out.bin: per frame 18 doubles
// {process() result, number of (bucketed) matches, Tr_delta row-major 4x4 after the frame}.
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "viso_stereo.h"

int main(int argc, char **argv) {
  if (argc < 7) return 2;
  FILE *f = fopen(argv[1], "rb");
  if (!f) return 3;
  int32_t hdr[3];
  if (fread(hdr, 4, 3, f) != 3) return 3;
  const int w = hdr[0], h = hdr[1], n = hdr[2];
  VisualOdometryStereo::parameters p;
  p.calib.f = atof(argv[3]);
  p.calib.cu = atof(argv[4]);
  p.calib.cv = atof(argv[5]);
  p.base = atof(argv[6]);
  VisualOdometryStereo vo(p);
  std::vector<uint8_t> L((size_t)w * h), R((size_t)w * h);
  FILE *o = fopen(argv[2], "wb");
  uint32_t dims[3] = {(uint32_t)w, (uint32_t)h, (uint32_t)w};
  for (int i = 0; i < n; i++) {
    if (fread(L.data(), 1, L.size(), f) != L.size() || fread(R.data(), 1, R.size(), f) != R.size()) return 4;
    bool ok = vo.process(L.data(), R.data(), dims);
    Matrix T = vo.getMotion();
    double rec[18];
    rec[0] = ok ? 1 : 0;
    rec[1] = (double)vo.getNumberOfMatches();
    for (int a = 0; a < 4; a++)
      for (int b = 0; b < 4; b++) rec[2 + a * 4 + b] = T.val[a][b];
    fwrite(rec, sizeof(double), 18, o);
  }
  fclose(o);
  fclose(f);
  return 0;
}
