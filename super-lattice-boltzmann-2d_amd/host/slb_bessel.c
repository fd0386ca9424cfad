/*
 * slb_bessel.c -- I_n(x) for integer order, used for the equilibrium
 * coefficients and the output multipliers (the reference calls GSL's
 * gsl_sf_bessel_In / gsl_sf_bessel_I0: src/boltzmann_solver.c:122, 359-360).
 *
 * Ascending series, all terms positive, carried in long double so the double
 * result is correctly rounded or one ulp off.  For n far above x the value is
 * below the double range and 0 is returned (its float image is 0 as well).
 */
#include "slb_host.h"

double slb_bessel_In(int n, double x)
{
  if (n < 0) n = -n;
  const long double half = 0.5L * (long double)x;
  const long double q = half * half;
  long double lead = 1.0L;                       /* (x/2)^n / n! */
  for (int i = 1; i <= n; i++) lead *= half / i;
  long double total = lead, t = lead;
  for (int k = 1; k < 200000; k++) {
    t *= q / ((long double)k * (long double)(k + n));
    total += t;
    if (t < total * 1e-23L) break;
  }
  return (double)total;
}
