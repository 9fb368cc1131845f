// numerics_check.c — bitwise compare of device dumps (experiments/numerics_probe.hip, part C)
// against the gcc build of dsm_numerics.h.  gcc -O2 -ffp-contract=off -mfma -mavx2
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "dsm_numerics.h"

static float* load(const char* dir, const char* name, long* n) {
  char path[512];
  snprintf(path, sizeof path, "%s/probe_num_%s.bin", dir, name);
  FILE* f = fopen(path, "rb");
  if (!f) { perror(path); exit(2); }
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  float* p = (float*)malloc(sz);
  if (fread(p, 1, sz, f) != (size_t)sz) exit(3);
  fclose(f);
  *n = sz / 4;
  return p;
}

int main(int argc, char** argv) {
  const char* dir = argc > 1 ? argv[1] : "gpurun_out";
  long n, m;
  float* x = load(dir, "x", &n);
  float* y = load(dir, "y", &m);
  const char* names[9] = {"exp", "elu", "silu", "gelu", "sin", "cos", "div", "rsqrt", "muladd"};
  int bad = 0;
  for (int k = 0; k < 9; ++k) {
    float* h = load(dir, names[k], &m);
    long mism = 0;
    double maxrel = 0;
    for (long i = 0; i < n; ++i) {
      float a = x[i], b = y[i], ref = 0, s, c;
      double truth = NAN;
      switch (k) {
        case 0: ref = dsm_expf(a); truth = exp((double)a); break;
        case 1: ref = dsm_elu(a); truth = a >= 0 ? a : expm1((double)a); break;
        case 2: ref = dsm_silu(a); truth = a / (1.0 + exp(-(double)a)); break;
        case 3: ref = dsm_gelu_erf(a); truth = 0.5 * a * (1.0 + erf(a / sqrt(2.0))); break;
        case 4: dsm_sincosf(fabsf(a) * 1000.0f, &s, &c); ref = s; truth = sin((double)(fabsf(a) * 1000.0f)); break;
        case 5: dsm_sincosf(fabsf(a) * 1000.0f, &s, &c); ref = c; truth = cos((double)(fabsf(a) * 1000.0f)); break;
        case 6: ref = a / b; break;
        case 7: ref = 1.0f / sqrtf(fabsf(b) + 1e-8f); break;
        case 8: { volatile float mm = a * b; ref = mm + a; } break;
      }
      if (memcmp(&ref, &h[i], 4) && !(ref != ref && h[i] != h[i])) {
        if (mism < 3) printf("   %s: x=%a y=%a cpu=%a gpu=%a\n", names[k], a, b, ref, h[i]);
        mism++;
      }
      if (truth == truth && isfinite(truth) && fabs(truth) > 1e-30 && isfinite(ref)) {
        double rel = fabs((double)ref - truth) / fabs(truth);
        if (k == 1 && a < 0 && a > -1e-3) rel = 0; /* exp(x)-1 cancellation is the reference's own formula */
        if (rel > maxrel) maxrel = rel;
      }
    }
    printf("%-7s gcc-vs-device mismatches: %ld / %ld   max rel err vs libm(double): %.3g\n", names[k], mism, n, maxrel);
    if (mism) bad = 1;
    free(h);
  }
  return bad;
}
