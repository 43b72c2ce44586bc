/*
 * tolg_oracle_ld.c -- the oracle in `long double` (x87 80-bit, 64-bit mantissa): a REFEREE for rounding-level
 * disagreements between the fp64 GPU path and the fp64 oracle.  TEST INFRASTRUCTURE ONLY, like tolg_oracle.c; used by
 * tools/parity_referee.py and tests/test_oracle_ld.py, never by the product and never as a parity target -- parity is
 * against tolg_oracle.c.  When two fp64 implementations of the same algorithm differ by 1e-9 after seven iterations, the
 * question is whether one of them lost precision somewhere or whether the problem amplifies the last bit of anything:
 * the same statements evaluated with 11 more mantissa bits answer it (both sides about equally far from this twin = the
 * problem; one side much farther = that side).
 *
 * Nothing is restated here: the Makefile derives _gen/tolg_oracle_ld_body.c from tolg_oracle.c by giving every floating
 * literal an `L` suffix, and this file compiles that body with `double` read as `long double` and <tgmath.h>'s
 * type-generic sin / cos / sqrt / atan2 / fabs.  The system headers are included first, under their own types; their
 * include guards keep the body's #include lines from expanding again under the macro.  The exported functions keep
 * their names and argument lists with long double in place of double (oracle/bridge_ld.py).
 */
#include <tgmath.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>
#define double long double
#include "_gen/tolg_oracle_ld_body.c"
