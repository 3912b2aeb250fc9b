// Dual numbers for the forward-mode linearisation of the residual (setupStateResidualMatrix with useAD = T,
// adjointUtils.F90:227-409, masterRoutines::block_res_state_d :1285-1393).
//
// v + d eps with eps^2 = 0: every arithmetic operation and elementary function carries the derivative of its result with respect to
// ONE seed direction.  The gather kernels of the residual are compiled a second time with this type in the place of `double`
// (kernels_ad.hip), so that what is differentiated is the very expression the finite-difference assembly evaluates.
// Where a function has a kink the branch taken is the one the reference's Tapenade-generated code takes
// (src/adjoint/outputForward/*.f90): abs: x >= 0 -> +; max(a, b): a unless a < b; min(a, b): a unless a > b;
// sqrt(0): derivative 0; x**y: y x**(y-1) dx where x > 0 (or x < 0 with an integer y), dx at x = 0, y = 1, else 0.
#ifndef ADFLOW_DUAL_H
#define ADFLOW_DUAL_H
#include <cmath>

#ifndef __HIPCC__
#ifndef __host__
#define __host__
#define __device__
#endif
#endif

struct Dual {
    double v, d;
    Dual() = default;
    __host__ __device__ Dual(double a) : v(a), d(0.0) {}
    __host__ __device__ Dual(double a, double b) : v(a), d(b) {}
};

__host__ __device__ inline Dual operator+(const Dual& a, const Dual& b) { return Dual(a.v + b.v, a.d + b.d); }
__host__ __device__ inline Dual operator-(const Dual& a, const Dual& b) { return Dual(a.v - b.v, a.d - b.d); }
__host__ __device__ inline Dual operator*(const Dual& a, const Dual& b) { return Dual(a.v * b.v, a.d * b.v + a.v * b.d); }
__host__ __device__ inline Dual operator/(const Dual& a, const Dual& b)
{
    const double q = a.v / b.v;
    return Dual(q, (a.d - q * b.d) / b.v);
}
__host__ __device__ inline Dual operator+(const Dual& a, double b) { return Dual(a.v + b, a.d); }
__host__ __device__ inline Dual operator+(double a, const Dual& b) { return Dual(a + b.v, b.d); }
__host__ __device__ inline Dual operator-(const Dual& a, double b) { return Dual(a.v - b, a.d); }
__host__ __device__ inline Dual operator-(double a, const Dual& b) { return Dual(a - b.v, -b.d); }
__host__ __device__ inline Dual operator*(const Dual& a, double b) { return Dual(a.v * b, a.d * b); }
__host__ __device__ inline Dual operator*(double a, const Dual& b) { return Dual(a * b.v, a * b.d); }
__host__ __device__ inline Dual operator/(const Dual& a, double b) { return Dual(a.v / b, a.d / b); }
__host__ __device__ inline Dual operator/(double a, const Dual& b)
{
    const double q = a / b.v;
    return Dual(q, -q * b.d / b.v);
}
__host__ __device__ inline Dual operator-(const Dual& a) { return Dual(-a.v, -a.d); }
__host__ __device__ inline Dual operator+(const Dual& a) { return a; }
__host__ __device__ inline Dual& operator+=(Dual& a, const Dual& b) { a.v += b.v; a.d += b.d; return a; }
__host__ __device__ inline Dual& operator-=(Dual& a, const Dual& b) { a.v -= b.v; a.d -= b.d; return a; }
__host__ __device__ inline Dual& operator*=(Dual& a, const Dual& b) { a = a * b; return a; }
__host__ __device__ inline Dual& operator/=(Dual& a, const Dual& b) { a = a / b; return a; }
__host__ __device__ inline Dual& operator+=(Dual& a, double b) { a.v += b; return a; }
__host__ __device__ inline Dual& operator-=(Dual& a, double b) { a.v -= b; return a; }
__host__ __device__ inline Dual& operator*=(Dual& a, double b) { a.v *= b; a.d *= b; return a; }
__host__ __device__ inline Dual& operator/=(Dual& a, double b) { a.v /= b; a.d /= b; return a; }

#define ADF_DUAL_CMP(OP)                                                                              \
    __host__ __device__ inline bool operator OP(const Dual& a, const Dual& b) { return a.v OP b.v; }  \
    __host__ __device__ inline bool operator OP(const Dual& a, double b) { return a.v OP b; }         \
    __host__ __device__ inline bool operator OP(double a, const Dual& b) { return a OP b.v; }
ADF_DUAL_CMP(<)
ADF_DUAL_CMP(>)
ADF_DUAL_CMP(<=)
ADF_DUAL_CMP(>=)
ADF_DUAL_CMP(==)
ADF_DUAL_CMP(!=)
#undef ADF_DUAL_CMP

__host__ __device__ inline Dual sqrt(const Dual& a)
{
    const double r = ::sqrt(a.v);
    return Dual(r, a.v == 0.0 ? 0.0 : a.d / (2.0 * r));
}
// SIGN(a, b): |a| with the sign of b
__host__ __device__ inline Dual copysign(const Dual& a, const Dual& b)
{
    const bool same = (::copysign(1.0, a.v) == ::copysign(1.0, b.v));
    return Dual(::copysign(a.v, b.v), same ? a.d : -a.d);
}
__host__ __device__ inline Dual fabs(const Dual& a) { return a.v >= 0.0 ? a : Dual(-a.v, -a.d); }
__host__ __device__ inline Dual fmax(const Dual& a, const Dual& b) { return (a.v < b.v) ? b : a; }
__host__ __device__ inline Dual fmax(const Dual& a, double b) { return (a.v < b) ? Dual(b) : a; }
__host__ __device__ inline Dual fmax(double a, const Dual& b) { return (a < b.v) ? b : Dual(a); }
__host__ __device__ inline Dual fmin(const Dual& a, const Dual& b) { return (a.v > b.v) ? b : a; }
__host__ __device__ inline Dual fmin(const Dual& a, double b) { return (a.v > b) ? Dual(b) : a; }
__host__ __device__ inline Dual fmin(double a, const Dual& b) { return (a > b.v) ? b : Dual(a); }
__host__ __device__ inline Dual exp(const Dual& a)
{
    const double e = ::exp(a.v);
    return Dual(e, e * a.d);
}
__host__ __device__ inline Dual log(const Dual& a) { return Dual(::log(a.v), a.d / a.v); }
__host__ __device__ inline Dual tanh(const Dual& a)
{
    const double t = ::tanh(a.v);
    return Dual(t, (1.0 - t * t) * a.d);
}
// x ** y with a passive exponent
__host__ __device__ inline Dual pow(const Dual& x, double y)
{
    const double r = ::pow(x.v, y);
    double d;
    if (x.v > 0.0 || (x.v < 0.0 && y == (double)(long)y)) d = y * ::pow(x.v, y - 1.0) * x.d;
    else if (x.v == 0.0 && y == 1.0) d = x.d;
    else d = 0.0;
    return Dual(r, d);
}
// both active (the spectral-radius scaling (ri / rj) ** adis has a passive exponent, SA's chi ** 3 etc. are products in the
// kernels; this form is here for completeness): d = y x**(y-1) dx + x**y log(x) dy, the second term where x > 0
__host__ __device__ inline Dual pow(const Dual& x, const Dual& y)
{
    Dual r = pow(x, y.v);
    if (x.v > 0.0) r.d += r.v * ::log(x.v) * y.d;
    return r;
}
__host__ __device__ inline Dual pow(double x, const Dual& y)
{
    const double r = ::pow(x, y.v);
    return Dual(r, x > 0.0 ? r * ::log(x) * y.d : 0.0);
}
__host__ __device__ inline bool isfinite(const Dual& a) { return ::isfinite(a.v) && ::isfinite(a.d); }

#endif
