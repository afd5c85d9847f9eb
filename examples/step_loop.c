/* The time loop of the reference's scripts/prob1.py:65-98 written against the
 * C ABI alone (plain C99, no C++ and no Python): channel flow on a 40 x 1
 * domain, 38 x 18 cells, implicit Euler with GMRESSolver<10>, dt = 0.25, until
 * the largest temperature update falls to 1e-9.
 *
 *   gcc -std=c99 -I include examples/step_loop.c -L krylov_amd -lkrylov_amd -Wl,-rpath,krylov_amd -lm
 *
 * Prints "steps <n> last_delta <d> sum_T <s>" (the unmodified reference:
 * 147 steps, sum of the interior temperatures 528.465360608386). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "krylov_amd.h"

#define CX 38
#define CY 18

static void die(const char *what) {
  fprintf(stderr, "%s: %s\n", what, kry_last_error());
  exit(1);
}

int main(int argc, char **argv) {
  const int m = argc > 1 ? atoi(argv[1]) : 10;
  const double dx = 40.0 / CX, dy = 1.0 / CY;
  static double T0[CX * CY], u0[CX * CY], v0[CX * CY], padded[(CX + 2) * (CY + 2)];
  double side[CX > CY ? CX : CY];
  /* T bottom/top/left Dirichlet, right Neumann; u bottom/top Dirichlet, left/right Neumann; v all Dirichlet */
  const int dirichlet[12] = {1, 1, 1, 0, 1, 1, 0, 0, 1, 1, 1, 1};
  kry_ctx *ctx = NULL;
  kry_ie *ie = NULL;
  double delta = INFINITY, sum = 0.0;
  int i, j, f, s, steps = 0;

  for (i = 0; i < CX; i++)
    for (j = 0; j < CY; j++) {
      const double y = 0.0 + dy / 2 + j * dy; /* Mesh::median_y */
      T0[i * CY + j] = 0.0;
      u0[i * CY + j] = 6.0 * 3.0 * y * (1.0 - y);
      v0[i * CY + j] = 0.0;
    }
  if (kry_ctx_create(0, NULL, &ctx) != KRY_OK) die("kry_ctx_create");
  if (kry_ie_create(ctx, 0.0, 0.0, 40.0, 1.0, CX, CY, T0, u0, v0, 1.0, 25.0, 0.7, 0.1, dirichlet, m, KRY_GMRES_PLAIN,
                    &ie) != KRY_OK)
    die("kry_ie_create");
  /* boundary values (constant in time here): everything 0 except T = 1 on top and T = y on the left */
  for (f = 0; f < 3; f++)
    for (s = 0; s < 4; s++) {
      const int len = (s == KRY_SIDE_BOTTOM || s == KRY_SIDE_TOP) ? CX : CY;
      for (i = 0; i < len; i++) {
        side[i] = 0.0;
        if (f == KRY_FIELD_T && s == KRY_SIDE_TOP) side[i] = 1.0;
        if (f == KRY_FIELD_T && s == KRY_SIDE_LEFT) side[i] = 0.0 + dy / 2 + i * dy;
      }
      if (kry_ie_set_bc_values(ie, f, s, side, len) != KRY_OK) die("kry_ie_set_bc_values");
    }
  if (kry_ie_init_offdiag(ie) != KRY_OK) die("kry_ie_init_offdiag");
  while (delta > 1e-9 && steps < 1000) {
    if (kry_ie_timestep(ie, 0.25, &delta) != KRY_OK) die("kry_ie_timestep");
    steps++;
  }
  if (kry_ie_get_field(ie, KRY_FIELD_T, padded) != KRY_OK) die("kry_ie_get_field");
  for (i = 1; i <= CX; i++)
    for (j = 1; j <= CY; j++) sum += padded[i * (CY + 2) + j];
  printf("steps %d last_delta %.17g sum_T %.15g time %.17g\n", steps, delta, sum, kry_ie_time(ie));
  (void)dx;
  kry_ie_destroy(ie);
  kry_ctx_destroy(ctx);
  return 0;
}
