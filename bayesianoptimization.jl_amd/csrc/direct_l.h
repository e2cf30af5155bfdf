// direct_l.h -- DIviding RECTangles, locally biased (Gablonsky & Kelley 2001): the HOST bookkeeping of the search the
// reference runs by default for ThompsonSamplingSimple (reference src/acquisition.jl:7-9: method = :GN_DIRECT_L, restarts = 1,
// maxeval = 2000; nlopt_setup :20-38 hands the acquisition to NLopt's GN_DIRECT_L).  NLopt is a dependency of the reference that is not
// vendored (Project.toml: NLopt); what is restated here is the published algorithm with the rules of NLopt's cdirect.c for this
// variant (its `which_alg = 13`):
//   * a rectangle's size is its LONGEST side (level l: side 3^-l);
//   * potentially optimal = upper convex hull of (size, best value of that size), from the incumbent's size class towards the larger
//     rectangles, ONE rectangle per size class (Jones' epsilon = 0; first index wins ties);
//   * a cube is trisected along EVERY side, best sampled value first (so the best points keep the largest boxes); any other rectangle
//     along its first longest side only.
// The search is MAXIMISATION and batched: ask() hands out ALL new centres of one iteration (one device call scores them), tell()
// takes their values.  What is not reproduced is NLopt's evaluation ORDER inside an iteration (irrelevant for a deterministic
// objective, a different random stream for a sampled one).
// Round 6 measured the same bookkeeping as NumPy code in the Python mirror at 21 of the 27.7 ms of a default Thompson acquire_max
// (123 iterations, 2000 points); acquisition._batched_direct_l stays as the twin the tests compare this against, bit for bit.
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

namespace bohip {

struct DirectL {
    int64_t d = 0, maxeval = 0;
    double stopval = INFINITY;
    std::vector<double> lb, span;
    std::vector<double> C;        // centres in the unit cube, d x n column-major
    std::vector<int64_t> Lv;      // level per side, d x n
    std::vector<double> F;        // value per rectangle (NaN -> -Inf)
    int64_t evals = 0, iterations = 0;
    bool started = false, finished = false;
    bool has_deadline = false;
    std::chrono::steady_clock::time_point deadline;
    // the batch handed out by ask() and not yet told
    std::vector<double> U;        // d x m, unit cube
    struct Plan { int64_t j; std::vector<int64_t> dims; int64_t start; };
    std::vector<Plan> plan;

    DirectL(int64_t d_, const double* lb_, const double* ub_, int64_t maxeval_, double stopval_, double maxtime)
        : d(d_), maxeval(std::max<int64_t>(1, maxeval_)), stopval(stopval_), lb(lb_, lb_ + d_), span(d_) {
        for (int64_t i = 0; i < d; ++i) span[i] = ub_[i] - lb_[i];
        if (maxtime > 0) {
            has_deadline = true;
            deadline = std::chrono::steady_clock::now() +
                       std::chrono::duration_cast<std::chrono::steady_clock::duration>(std::chrono::duration<double>(maxtime));
        }
    }
    int64_t n() const { return (int64_t)F.size(); }
    int64_t pending() const { return d ? (int64_t)U.size() / d : 0; }

    void to_x(const double* u, double* x) const {
#pragma clang fp contract(off)
        for (int64_t i = 0; i < d; ++i) {
            const double t = span[i] * u[i];
            x[i] = lb[i] + t;
        }
    }
    int64_t argmax_first() const {
        int64_t jb = 0;
        for (int64_t j = 1; j < n(); ++j)
            if (F[j] > F[jb]) jb = j;
        return jb;
    }

    // Plans the next iteration; returns the number of new points (0 = the search is over).
    int64_t plan_next() {
#pragma clang fp contract(off)
        if (!U.empty()) return pending();                      // asked twice without a tell: the same batch again
        if (finished) return 0;
        if (!started) {                                        // the box's centre
            U.assign(d, 0.5);
            return 1;
        }
        const int64_t nr = n();
        const int64_t jmax = argmax_first();
        if (!(evals < maxeval) || F[jmax] >= stopval ||
            (has_deadline && !(std::chrono::steady_clock::now() < deadline))) {
            finished = true;
            return 0;
        }
        // size class of every rectangle = its smallest level; per class the best rectangle, first on ties
        std::vector<int64_t> size(nr);
        int64_t kmax = 0;
        for (int64_t j = 0; j < nr; ++j) {
            int64_t m = Lv[j * d];
            for (int64_t i = 1; i < d; ++i) m = std::min(m, Lv[j * d + i]);
            size[j] = m;
            kmax = std::max(kmax, m);
        }
        std::vector<int64_t> best_of(kmax + 1, -1);
        for (int64_t j = 0; j < nr; ++j) {
            int64_t& b = best_of[size[j]];
            if (b < 0 || F[j] > F[b]) b = j;
        }
        // upper hull over (diameter, value) from the incumbent's class towards the larger rectangles (decreasing level)
        struct Pt { double x, y; int64_t j; };
        std::vector<Pt> hull;
        for (int64_t k = size[jmax]; k >= 0; --k) {
            if (best_of[k] < 0) continue;
            const Pt pt{std::pow(3.0, (double)(-k)), F[best_of[k]], best_of[k]};
            while (hull.size() >= 2) {
                const Pt &p1 = hull[hull.size() - 2], &p2 = hull[hull.size() - 1];
                if ((p2.y - p1.y) * (pt.x - p1.x) <= (pt.y - p1.y) * (p2.x - p1.x))   // the middle point is not above the chord
                    hull.pop_back();
                else
                    break;
            }
            hull.push_back(pt);
        }
        plan.clear();
        int64_t ncols = 0;
        for (const Pt& h : hull) {
            const int64_t j = h.j;
            const int64_t* lv = &Lv[j * d];
            const int64_t kmin = size[j];
            std::vector<int64_t> dims;
            for (int64_t i = 0; i < d; ++i)
                if (lv[i] == kmin) dims.push_back(i);
            if ((int64_t)dims.size() != d) dims.resize(1);     // not a cube: the first longest side only
            if (evals + ncols + 2 * (int64_t)dims.size() > maxeval)
                dims.resize((size_t)std::max<int64_t>(0, (maxeval - evals - ncols) / 2));
            if (dims.empty()) continue;
            const double delta = std::pow(3.0, (double)(-(kmin + 1)));
            const int64_t start = ncols;
            for (int64_t i : dims)
                for (int s = 0; s < 2; ++s) {
                    const size_t o = U.size();
                    U.insert(U.end(), &C[j * d], &C[j * d] + d);
                    U[o + i] += s == 0 ? delta : -delta;
                    ++ncols;
                }
            plan.push_back(Plan{j, std::move(dims), start});
        }
        if (ncols == 0) finished = true;
        return ncols;
    }

    // X: d x cap column-major, box coordinates.  Returns the number of points, 0 when the search is over, -1 when cap is too small.
    int64_t ask(double* X, int64_t cap) {
        const int64_t m = plan_next();
        if (m > cap) return -1;
        for (int64_t c = 0; c < m; ++c) to_x(&U[c * d], X + c * d);
        return m;
    }

    // Values of the batch ask() handed out, in its order.  Returns false when m is not that batch's size.
    bool tell(const double* Fnew_, int64_t m) {
        if (m != pending() || m == 0) return false;
        std::vector<double> Fnew(Fnew_, Fnew_ + m);
        for (double& f : Fnew)
            if (std::isnan(f)) f = -INFINITY;
        if (!started) {
            started = true;
            C = U;
            Lv.assign(d, 0);
            F = Fnew;
            evals = 1;
            U.clear();
            return true;
        }
        const int64_t n0 = n();
        evals += m;
        ++iterations;
        Lv.resize((size_t)(n0 + m) * d);
        for (const Plan& p : plan) {
            const int64_t nd = (int64_t)p.dims.size();
            std::vector<double> w(nd);
            for (int64_t t = 0; t < nd; ++t) w[t] = std::max(Fnew[p.start + 2 * t], Fnew[p.start + 2 * t + 1]);
            std::vector<int64_t> order(nd);
            for (int64_t t = 0; t < nd; ++t) order[t] = t;
            std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return w[a] > w[b]; });   // best sampled value first
            int64_t* lv = &Lv[p.j * d];
            for (int64_t t : order) {
                lv[p.dims[t]] += 1;          // the parent shrinks along this side; the two children inherit the levels so far
                std::copy(lv, lv + d, &Lv[(n0 + p.start + 2 * t) * d]);
                std::copy(lv, lv + d, &Lv[(n0 + p.start + 2 * t + 1) * d]);
            }
        }
        C.insert(C.end(), U.begin(), U.end());
        F.insert(F.end(), Fnew.begin(), Fnew.end());
        U.clear();
        plan.clear();
        return true;
    }

    // best value so far (first maximum), its point in box coordinates
    double best(double* x) const {
        if (F.empty()) {                       // nothing told yet: the box's centre, as the first ask would hand out
            if (x) { const std::vector<double> c(d, 0.5); to_x(c.data(), x); }
            return -INFINITY;
        }
        const int64_t jb = argmax_first();
        if (x) to_x(&C[jb * d], x);
        return F[jb];
    }
};

}  // namespace bohip
