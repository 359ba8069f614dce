// lmfit.hpp -- HOST code: the Levenberg-Marquardt fit behind the reference's edge maps, reproduced to the last bit.
//
// utils/keypoint2img.py interp_points (:319-354; the pose variant utils/keypoint2img_posenorm.py :490-516 is the same function) fits every
// 2- or 3-point piece of a key-point polyline with scipy.optimize.curve_fit(f, x, y) -- f = a x + b or a x^2 + b x + c, no bounds, no
// Jacobian, p0 = ones -- and TRUNCATES the sampled curve to integer pixels.  The fitted ordinate at an integer key point is 217.999.. or
// 218.000.. by the optimiser's last bits, so which pixel is set depends on the exact floating-point path of the fit: a closed-form
// parabola (round 2) differed from the reference on 2.5 % of the edge pixels.  curve_fit with method "lm" is scipy.optimize.leastsq ->
// MINPACK lmdif (pinned third-party arithmetic: scipy 1.15.3 here, C translation of the Fortran original, same algorithm) with
// ftol = xtol = 1.49012e-8, gtol = 0, maxfev = 200 (n + 1), epsfcn = machine epsilon, factor = 100, mode 1 (internal scaling).
// This is that algorithm (lmdif, fdjac2, qrfac, lmpar, qrsolv, enorm) for m, n <= 3, every floating-point operation in MINPACK's order and
// without contraction (-ffp-contract=off): tests/test_raster.py checks the coefficients against curve_fit bit for bit on every piece of
// the reference's demo clips (2652 + 7080 fits), and the device maps drawn from them equal the reference's maps.
// Residuals are evaluated as numpy evaluates `f(x, *p) - y`: ((a * (x * x)) + (b * x)) + c, then - y; (a * x) + b, then - y.
#pragma once
#include <cmath>

namespace tsnet {
namespace lm {

constexpr double kEpsMch = 2.220446049250313e-16;          // dpmpar(1)
constexpr double kDwarf = 2.2250738585072014e-308;         // dpmpar(2)
constexpr int MAXN = 3;

inline double enorm(int n, const double* x) {
    const double rdwarf = 3.834e-20, rgiant = 1.304e19;
    double s1 = 0, s2 = 0, s3 = 0, x1max = 0, x3max = 0;
    if (n <= 0) return 0.0;
    const double agiant = rgiant / (double)n;
    for (int i = 0; i < n; ++i) {
        const double xabs = std::fabs(x[i]);
        if (xabs > rdwarf && xabs < agiant) {
            s2 += xabs * xabs;
        } else if (xabs <= rdwarf) {
            if (xabs > x3max) { const double t = x3max / xabs; s3 = 1.0 + s3 * (t * t); x3max = xabs; }
            else if (xabs != 0.0) { const double t = xabs / x3max; s3 += t * t; }
        } else {
            if (xabs > x1max) { const double t = x1max / xabs; s1 = 1.0 + s1 * (t * t); x1max = xabs; }
            else { const double t = xabs / x1max; s1 += t * t; }
        }
    }
    if (s1 != 0.0) return x1max * std::sqrt(s1 + (s2 / x1max) / x1max);
    if (s2 != 0.0) {
        if (s2 >= x3max) return std::sqrt(s2 * (1.0 + (x3max / s2) * (x3max * s3)));
        return std::sqrt(x3max * ((s2 / x3max) + (x3max * s3)));
    }
    return x3max * std::sqrt(s3);
}

// the model: quadratic (n = 3) or linear (n = 2) through m points; residual = f(x, p) - y
struct Problem {
    int m, n;
    double x[MAXN], y[MAXN];
    void eval(const double* p, double* r) const {
        for (int i = 0; i < m; ++i) {
            double f;
            if (n == 3) f = ((p[0] * (x[i] * x[i])) + (p[1] * x[i])) + p[2];
            else f = (p[0] * x[i]) + p[1];
            r[i] = f - y[i];
        }
    }
};

// a[i][j]: row i, column j
inline void qrfac(int m, int n, double a[MAXN][MAXN], int* ipvt, double* rdiag, double* acnorm) {
    double wa[MAXN], col[MAXN];
    for (int j = 0; j < n; ++j) {
        for (int i = 0; i < m; ++i) col[i] = a[i][j];
        acnorm[j] = enorm(m, col);
        rdiag[j] = acnorm[j]; wa[j] = rdiag[j]; ipvt[j] = j;
    }
    const int minmn = m < n ? m : n;
    for (int j = 0; j < minmn; ++j) {
        int kmax = j;
        for (int k = j; k < n; ++k) if (rdiag[k] > rdiag[kmax]) kmax = k;
        if (kmax != j) {
            for (int i = 0; i < m; ++i) { const double t = a[i][j]; a[i][j] = a[i][kmax]; a[i][kmax] = t; }
            rdiag[kmax] = rdiag[j]; wa[kmax] = wa[j];
            const int t = ipvt[j]; ipvt[j] = ipvt[kmax]; ipvt[kmax] = t;
        }
        for (int i = j; i < m; ++i) col[i - j] = a[i][j];
        double ajnorm = enorm(m - j, col);
        if (ajnorm != 0.0) {
            if (a[j][j] < 0.0) ajnorm = -ajnorm;
            for (int i = j; i < m; ++i) a[i][j] /= ajnorm;
            a[j][j] += 1.0;
            for (int k = j + 1; k < n; ++k) {
                double s = 0.0;
                for (int i = j; i < m; ++i) s += a[i][j] * a[i][k];
                double temp = s / a[j][j];
                for (int i = j; i < m; ++i) a[i][k] -= temp * a[i][j];
                if (rdiag[k] != 0.0) {
                    temp = a[j][k] / rdiag[k];
                    const double d = 1.0 - temp * temp;
                    rdiag[k] *= std::sqrt(d > 0.0 ? d : 0.0);
                    const double t2 = rdiag[k] / wa[k];
                    if (0.05 * (t2 * t2) <= kEpsMch) {
                        for (int i = j + 1; i < m; ++i) col[i - j - 1] = a[i][k];
                        rdiag[k] = enorm(m - j - 1, col);
                        wa[k] = rdiag[k];
                    }
                }
            }
        }
        rdiag[j] = -ajnorm;
    }
}

inline void qrsolv(int n, double r[MAXN][MAXN], const int* ipvt, const double* diag, const double* qtb, double* x, double* sdiag) {
    double wa[MAXN];
    for (int j = 0; j < n; ++j) {
        for (int i = j; i < n; ++i) r[i][j] = r[j][i];
        x[j] = r[j][j]; wa[j] = qtb[j];
    }
    for (int j = 0; j < n; ++j) {
        const int l = ipvt[j];
        if (diag[l] != 0.0) {
            for (int k = j; k < n; ++k) sdiag[k] = 0.0;
            sdiag[j] = diag[l];
            double qtbpj = 0.0;
            for (int k = j; k < n; ++k) {
                if (sdiag[k] == 0.0) continue;
                double sn, cs;
                if (std::fabs(r[k][k]) < std::fabs(sdiag[k])) {
                    const double cotan = r[k][k] / sdiag[k];
                    sn = 0.5 / std::sqrt(0.25 + 0.25 * (cotan * cotan));
                    cs = sn * cotan;
                } else {
                    const double tn = sdiag[k] / r[k][k];
                    cs = 0.5 / std::sqrt(0.25 + 0.25 * (tn * tn));
                    sn = cs * tn;
                }
                r[k][k] = cs * r[k][k] + sn * sdiag[k];
                double temp = cs * wa[k] + sn * qtbpj;
                qtbpj = -sn * wa[k] + cs * qtbpj;
                wa[k] = temp;
                for (int i = k + 1; i < n; ++i) {
                    temp = cs * r[i][k] + sn * sdiag[i];
                    sdiag[i] = -sn * r[i][k] + cs * sdiag[i];
                    r[i][k] = temp;
                }
            }
        }
        sdiag[j] = r[j][j]; r[j][j] = x[j];
    }
    int nsing = n;
    for (int j = 0; j < n; ++j) {
        if (sdiag[j] == 0.0 && nsing == n) nsing = j;
        if (nsing < n) wa[j] = 0.0;
    }
    for (int k = 1; k <= nsing; ++k) {
        const int j = nsing - k;
        double s = 0.0;
        for (int i = j + 1; i < nsing; ++i) s += r[i][j] * wa[i];
        wa[j] = (wa[j] - s) / sdiag[j];
    }
    for (int j = 0; j < n; ++j) x[ipvt[j]] = wa[j];
}

inline double lmpar(int n, double r[MAXN][MAXN], const int* ipvt, const double* diag, const double* qtb, double delta, double par,
                    double* x, double* sdiag) {
    double wa1[MAXN], wa2[MAXN];
    int nsing = n;
    for (int j = 0; j < n; ++j) {
        wa1[j] = qtb[j];
        if (r[j][j] == 0.0 && nsing == n) nsing = j;
        if (nsing < n) wa1[j] = 0.0;
    }
    for (int k = 1; k <= nsing; ++k) {
        const int j = nsing - k;
        wa1[j] /= r[j][j];
        const double temp = wa1[j];
        for (int i = 0; i < j; ++i) wa1[i] -= r[i][j] * temp;
    }
    for (int j = 0; j < n; ++j) x[ipvt[j]] = wa1[j];
    int iter = 0;
    for (int j = 0; j < n; ++j) wa2[j] = diag[j] * x[j];
    double dxnorm = enorm(n, wa2);
    double fp = dxnorm - delta;
    if (fp <= 0.1 * delta) return 0.0;
    double parl = 0.0;
    if (nsing >= n) {
        for (int j = 0; j < n; ++j) { const int l = ipvt[j]; wa1[j] = diag[l] * (wa2[l] / dxnorm); }
        for (int j = 0; j < n; ++j) {
            double s = 0.0;
            for (int i = 0; i < j; ++i) s += r[i][j] * wa1[i];
            wa1[j] = (wa1[j] - s) / r[j][j];
        }
        const double temp = enorm(n, wa1);
        parl = ((fp / delta) / temp) / temp;
    }
    for (int j = 0; j < n; ++j) {
        double s = 0.0;
        for (int i = 0; i <= j; ++i) s += r[i][j] * qtb[i];
        wa1[j] = s / diag[ipvt[j]];
    }
    const double gnorm = enorm(n, wa1);
    double paru = gnorm / delta;
    if (paru == 0.0) paru = kDwarf / (delta < 0.1 ? delta : 0.1);
    par = par > parl ? par : parl;
    par = par < paru ? par : paru;
    if (par == 0.0) par = gnorm / dxnorm;
    for (;;) {
        ++iter;
        if (par == 0.0) { const double t = 0.001 * paru; par = kDwarf > t ? kDwarf : t; }
        double temp = std::sqrt(par);
        for (int j = 0; j < n; ++j) wa1[j] = temp * diag[j];
        qrsolv(n, r, ipvt, wa1, qtb, x, sdiag);
        for (int j = 0; j < n; ++j) wa2[j] = diag[j] * x[j];
        dxnorm = enorm(n, wa2);
        temp = fp;
        fp = dxnorm - delta;
        if (std::fabs(fp) <= 0.1 * delta || (parl == 0.0 && fp <= temp && temp < 0.0) || iter == 10) break;
        for (int j = 0; j < n; ++j) { const int l = ipvt[j]; wa1[j] = diag[l] * (wa2[l] / dxnorm); }
        for (int j = 0; j < n; ++j) {
            wa1[j] /= sdiag[j];
            const double t = wa1[j];
            for (int i = j + 1; i < n; ++i) wa1[i] -= r[i][j] * t;
        }
        temp = enorm(n, wa1);
        const double parc = ((fp / delta) / temp) / temp;
        if (fp > 0.0) parl = parl > par ? parl : par;
        if (fp < 0.0) paru = paru < par ? paru : par;
        par = parl > par + parc ? parl : par + parc;
    }
    return par;
}

// scipy.optimize.leastsq(func, ones(n)) for `prob`; returns MINPACK's info (1..4 = converged: what curve_fit accepts)
inline int lmdif(const Problem& prob, double* x) {
    const int m = prob.m, n = prob.n;
    const double ftol = 1.49012e-8, xtol = 1.49012e-8, gtol = 0.0, factor = 100.0;
    const int maxfev = 200 * (n + 1);
    double fvec[MAXN], diag[MAXN], qtf[MAXN], wa1[MAXN], wa2[MAXN], wa3[MAXN], wa4[MAXN], fjac[MAXN][MAXN];
    int ipvt[MAXN];
    for (int j = 0; j < n; ++j) x[j] = 1.0;
    prob.eval(x, fvec);
    int nfev = 1;
    double fnorm = enorm(m, fvec);
    double par = 0.0, xnorm = 0.0, delta = 0.0, gnorm = 0.0;
    int iter = 1, info = 0;
    const double eps = std::sqrt(kEpsMch);          // sqrt(max(epsfcn, epsmch)), epsfcn = machine epsilon
    for (;;) {
        for (int j = 0; j < n; ++j) {                // fdjac2: forward differences
            const double temp = x[j];
            double h = eps * std::fabs(temp);
            if (h == 0.0) h = eps;
            x[j] = temp + h;
            prob.eval(x, wa4);
            x[j] = temp;
            for (int i = 0; i < m; ++i) fjac[i][j] = (wa4[i] - fvec[i]) / h;
        }
        nfev += n;
        qrfac(m, n, fjac, ipvt, wa1, wa2);
        if (iter == 1) {
            for (int j = 0; j < n; ++j) { diag[j] = wa2[j]; if (wa2[j] == 0.0) diag[j] = 1.0; }
            for (int j = 0; j < n; ++j) wa3[j] = diag[j] * x[j];
            xnorm = enorm(n, wa3);
            delta = factor * xnorm;
            if (delta == 0.0) delta = factor;
        }
        for (int i = 0; i < m; ++i) wa4[i] = fvec[i];
        for (int j = 0; j < n; ++j) {
            if (fjac[j][j] != 0.0) {
                double s = 0.0;
                for (int i = j; i < m; ++i) s += fjac[i][j] * wa4[i];
                const double temp = -s / fjac[j][j];
                for (int i = j; i < m; ++i) wa4[i] += fjac[i][j] * temp;
            }
            fjac[j][j] = wa1[j];
            qtf[j] = wa4[j];
        }
        gnorm = 0.0;
        if (fnorm != 0.0) {
            for (int j = 0; j < n; ++j) {
                const int l = ipvt[j];
                if (wa2[l] == 0.0) continue;
                double s = 0.0;
                for (int i = 0; i <= j; ++i) s += fjac[i][j] * (qtf[i] / fnorm);
                const double g = std::fabs(s / wa2[l]);
                gnorm = gnorm > g ? gnorm : g;
            }
        }
        if (gnorm <= gtol) { info = 4; break; }
        for (int j = 0; j < n; ++j) diag[j] = diag[j] > wa2[j] ? diag[j] : wa2[j];
        for (;;) {
            double p[MAXN], sdiag[MAXN];
            par = lmpar(n, fjac, ipvt, diag, qtf, delta, par, p, sdiag);
            for (int j = 0; j < n; ++j) { wa1[j] = -p[j]; wa2[j] = x[j] + wa1[j]; wa3[j] = diag[j] * wa1[j]; }
            const double pnorm = enorm(n, wa3);
            if (iter == 1) delta = delta < pnorm ? delta : pnorm;
            prob.eval(wa2, wa4);
            ++nfev;
            const double fnorm1 = enorm(m, wa4);
            double actred = -1.0;
            if (0.1 * fnorm1 < fnorm) { const double t = fnorm1 / fnorm; actred = 1.0 - t * t; }
            for (int j = 0; j < n; ++j) wa3[j] = 0.0;
            for (int j = 0; j < n; ++j) {
                const double temp = wa1[ipvt[j]];
                for (int i = 0; i <= j; ++i) wa3[i] += fjac[i][j] * temp;
            }
            const double temp1 = enorm(n, wa3) / fnorm;
            const double temp2 = (std::sqrt(par) * pnorm) / fnorm;
            const double prered = temp1 * temp1 + (temp2 * temp2) / 0.5;
            const double dirder = -(temp1 * temp1 + temp2 * temp2);
            double ratio = 0.0;
            if (prered != 0.0) ratio = actred / prered;
            if (ratio <= 0.25) {
                double temp = 0.5;
                if (actred < 0.0) temp = 0.5 * dirder / (dirder + 0.5 * actred);
                if (0.1 * fnorm1 >= fnorm || temp < 0.1) temp = 0.1;
                const double pn = pnorm / 0.1;
                delta = temp * (delta < pn ? delta : pn);
                par = par / temp;
            } else if (par == 0.0 || ratio >= 0.75) {
                delta = pnorm / 0.5;
                par = 0.5 * par;
            }
            if (ratio >= 1e-4) {
                for (int j = 0; j < n; ++j) { x[j] = wa2[j]; wa2[j] = diag[j] * x[j]; }
                for (int i = 0; i < m; ++i) fvec[i] = wa4[i];
                xnorm = enorm(n, wa2);
                fnorm = fnorm1;
                ++iter;
            }
            const bool small = std::fabs(actred) <= ftol && prered <= ftol && 0.5 * ratio <= 1.0;
            if (small) info = 1;
            if (delta <= xtol * xnorm) info = 2;
            if (small && info == 2) info = 3;
            if (info != 0) break;
            if (nfev >= maxfev) info = 5;
            if (std::fabs(actred) <= kEpsMch && prered <= kEpsMch && 0.5 * ratio <= 1.0) info = 6;
            if (delta <= kEpsMch * xnorm) info = 7;
            if (gnorm <= kEpsMch) info = 8;
            if (info != 0) break;
            if (ratio >= 1e-4) break;
        }
        if (info != 0) break;
    }
    return info;
}

// One piece of interp_points: npts (2 or 3) points (x, y).  rec[8] = {kind, a, b, c, u_first, u_last, 0, 0}: kind 0 = nothing is drawn
// (|a| > 1, :333-334; or the fit did not converge: curve_fit raises there); bit 0 drawn, bit 1 fitted along y (the axis with the larger
// extent, :320-321: the curve is x(y)), bit 2 quadratic.  u_first <= u_last: the sample range runs upwards (:335-337).
inline void fit_piece(const double* pts, int npts, double* rec) {
    for (int i = 0; i < 8; ++i) rec[i] = 0.0;
    if (npts != 2 && npts != 3) return;
    double mdx = 0.0, mdy = 0.0;
    for (int i = 0; i + 1 < npts; ++i) {
        const double dx = std::fabs(pts[2 * i] - pts[2 * i + 2]), dy = std::fabs(pts[2 * i + 1] - pts[2 * i + 3]);
        mdx = dx > mdx ? dx : mdx; mdy = dy > mdy ? dy : mdy;
    }
    const bool swap = mdx < mdy;
    Problem pr; pr.m = npts; pr.n = npts;
    for (int i = 0; i < npts; ++i) { pr.x[i] = swap ? pts[2 * i + 1] : pts[2 * i]; pr.y[i] = swap ? pts[2 * i] : pts[2 * i + 1]; }
    for (int i = 0; i < npts; ++i) if (!std::isfinite(pr.x[i]) || !std::isfinite(pr.y[i])) return;     // curve_fit: asarray_chkfinite raises
    double p[MAXN] = {0, 0, 0};
    const int info = lmdif(pr, p);
    if (info < 1 || info > 4) return;
    if (npts == 3 && std::fabs(p[0]) > 1.0) return;
    double u0 = pr.x[0], u1 = pr.x[npts - 1];
    if (u0 > u1) { const double t = u0; u0 = u1; u1 = t; }
    rec[0] = 1.0 + (swap ? 2.0 : 0.0) + (npts == 3 ? 4.0 : 0.0);
    if (npts == 3) { rec[1] = p[0]; rec[2] = p[1]; rec[3] = p[2]; }
    else { rec[1] = 0.0; rec[2] = p[0]; rec[3] = p[1]; }
    rec[4] = u0; rec[5] = u1;
}

}  // namespace lm
}  // namespace tsnet
