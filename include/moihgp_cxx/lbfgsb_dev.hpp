// lbfgsb_dev.hpp -- the projected L-BFGS of lbfgsb.hpp with every parameter-sized vector ON THE DEVICE, and the online learner on top of it.
//
// Why: the learners' outer loop (reference moihgp/include/moihgp/moihgp_online.h:40-72 under LBFGS++, LBFGSB.h:117-241) evaluates
// MOIHGP::update + the window objective per line-search point and then does ~40 dot products / axpys over vectors of
// M L + L + 1 + L P doubles -- 134 MB each at M = L = 4096 (BASELINE.json configs[2]).  moihgp_update_dev / moihgp_window_eval_dev keep
// the objective on the device; this header keeps theta, the gradient, the search direction and the m correction pairs there too, on the
// vector kernels of libmoihgp.so (include/moihgp.h "device vectors", csrc/vecops.hip).  Only scalars cross PCIe.
//
// Same algorithm, stopping rules and parameter struct as lbfgsb.hpp (gradient projection for the active set, an L-BFGS direction on the
// free variables, projected Armijo backtracking capped at max_step); the iterates equal the host solver's up to the rounding order of
// the reductions.  No HIP header is needed to use it: device memory comes from moihgp_dvec_alloc.
#ifndef MOIHGP_CXX_LBFGSB_DEV_HPP_
#define MOIHGP_CXX_LBFGSB_DEV_HPP_

#include <algorithm>
#include <cmath>
#include <limits>
#include <list>
#include <memory>
#include <chrono>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

extern "C" {
#include "../moihgp.h"
}
#include "lbfgsb.hpp"
#include "moihgp.hpp"

namespace moihgp {
namespace opt {

// Opt-in wall-clock accounting of the learner's phases (tools/cxx/learner_bench.cpp prints it): off unless phases().enabled is set.
struct Phases {
    bool enabled = false;
    std::map<std::string, double> seconds;
    std::map<std::string, long> calls;
};
inline Phases& phases() { static Phases p; return p; }
class PhaseTimer {
public:
    explicit PhaseTimer(const char* name) : m_name(name), m_on(phases().enabled) { if (m_on) m_t0 = std::chrono::steady_clock::now(); }
    ~PhaseTimer() {
        if (!m_on) return;
        phases().seconds[m_name] += std::chrono::duration<double>(std::chrono::steady_clock::now() - m_t0).count();
        phases().calls[m_name]++;
    }
private:
    const char* m_name;
    bool m_on;
    std::chrono::steady_clock::time_point m_t0;
};

inline void dv_check(int rc, const char* what) {
    if (rc != 0) throw std::runtime_error(std::string(what) + ": " + moihgp_last_error());
}

// n doubles of device memory (owned)
class DVec {
public:
    DVec() : m_p(nullptr), m_n(0) {}
    explicit DVec(size_t n) : m_p(nullptr), m_n(0) { resize(n); }
    ~DVec() { moihgp_dvec_free(m_p); }
    DVec(const DVec&) = delete;
    DVec& operator=(const DVec&) = delete;
    DVec(DVec&& o) noexcept : m_p(o.m_p), m_n(o.m_n) { o.m_p = nullptr; o.m_n = 0; }
    DVec& operator=(DVec&& o) noexcept { if (this != &o) { moihgp_dvec_free(m_p); m_p = o.m_p; m_n = o.m_n; o.m_p = nullptr; o.m_n = 0; } return *this; }
    void resize(size_t n) {
        if (n == m_n) return;
        moihgp_dvec_free(m_p);
        m_p = moihgp_dvec_alloc(n);
        if (!m_p) throw std::runtime_error(std::string("moihgp_dvec_alloc: ") + moihgp_last_error());
        m_n = n;
    }
    double* data() { return m_p; }
    const double* data() const { return m_p; }
    size_t size() const { return m_n; }
private:
    double* m_p;
    size_t m_n;
};

// The vector operations of the solver on one context (stream, reduction scratch)
class DevOps {
public:
    DevOps() : m_c(moihgp_dvec_ctx_new()) { if (!m_c) throw std::runtime_error(std::string("moihgp_dvec_ctx_new: ") + moihgp_last_error()); }
    ~DevOps() { moihgp_dvec_ctx_del(m_c); }
    DevOps(const DevOps&) = delete;
    DevOps& operator=(const DevOps&) = delete;
    moihgp_dvec_ctx* ctx() { return m_c; }
    double dot(const DVec& a, const DVec& b, const unsigned char* mask = nullptr) { double r; dv_check(moihgp_dvec_dot(m_c, a.size(), a.data(), b.data(), mask, &r), "dvec_dot"); return r; }
    void axpy(double alpha, const DVec& x, DVec& y, const unsigned char* mask = nullptr) { dv_check(moihgp_dvec_axpy(m_c, x.size(), alpha, x.data(), y.data(), mask), "dvec_axpy"); }
    void scale(double alpha, const DVec& x, DVec& y, const unsigned char* mask = nullptr) { dv_check(moihgp_dvec_scale(m_c, x.size(), alpha, x.data(), y.data(), mask), "dvec_scale"); }
    void sub(const DVec& a, const DVec& b, DVec& out) { dv_check(moihgp_dvec_sub(m_c, a.size(), a.data(), b.data(), out.data()), "dvec_sub"); }
    void copy(const DVec& src, DVec& dst) { dv_check(moihgp_dvec_copy(m_c, dst.data(), src.data(), src.size()), "dvec_copy"); }
    void upload(const std::vector<double>& h, DVec& d) { d.resize(h.size()); dv_check(moihgp_dvec_upload(m_c, d.data(), h.data(), h.size()), "dvec_upload"); }
    void download(const DVec& d, std::vector<double>& h) { h.resize(d.size()); dv_check(moihgp_dvec_download(m_c, h.data(), d.data(), d.size()), "dvec_download"); }
    void sync() { dv_check(moihgp_dvec_sync(m_c), "dvec_sync"); }
    void* stream() { return moihgp_dvec_ctx_stream(m_c); }          // hipStream_t of the context: what the handle's `_on` entries order themselves by
private:
    moihgp_dvec_ctx* m_c;
};

// BFGSMat of lbfgsb.hpp (LBFGSpp/BFGSMat.h:37-178) over device vectors.  Copies share the pairs (the learner hands last solve's matrix to
// the next objective, moihgp_online.h:182: 2 m parameter-sized vectors that nobody wants copied).
class DevBFGSMat {
public:
    DevBFGSMat() : m_m(0), m_theta(1.0), m_ncorr(0), m_ptr(0), m_ops(nullptr) {}
    void reset(DevOps* ops, size_t n, int m) {
        m_ops = ops; m_m = m; m_theta = 1.0; m_ncorr = 0; m_ptr = m;
        m_s.reset(new std::vector<DVec>(m)); m_y.reset(new std::vector<DVec>(m));
        for (int i = 0; i < m; i++) { (*m_s)[i].resize(n); (*m_y)[i].resize(n); }
        m_ys.assign(m, 0.0); m_alpha.assign(m, 0.0);
    }
    void add_correction(const DVec& s, const DVec& y) {              // BFGSMat.h:81-99
        const int loc = m_ptr % m_m;
        m_ops->copy(s, (*m_s)[loc]); m_ops->copy(y, (*m_y)[loc]);
        const double ys = m_ops->dot(s, y);
        m_ys[loc] = ys;
        m_theta = m_ops->dot(y, y) / ys;
        if (m_ncorr < m_m) m_ncorr++;
        m_ptr = loc + 1;
    }
    // res = a H v, two-loop recursion with H0 = I / theta (BFGSMat.h:151-178)
    void apply_Hv(const DVec& v, const double& a, DVec& res) {
        res.resize(v.size());
        m_ops->scale(a, v, res);
        int j = m_ptr % m_m;
        for (int i = 0; i < m_ncorr; i++) {
            j = (j + m_m - 1) % m_m;
            m_alpha[j] = m_ops->dot((*m_s)[j], res) / m_ys[j];
            m_ops->axpy(-m_alpha[j], (*m_y)[j], res);
        }
        m_ops->scale(1.0 / m_theta, res, res);
        for (int i = 0; i < m_ncorr; i++) {
            const double beta = m_ops->dot((*m_y)[j], res) / m_ys[j];
            m_ops->axpy(m_alpha[j] - beta, (*m_s)[j], res);
            j = (j + 1) % m_m;
        }
    }
    // the same recursion with every pair restricted to the free variables (the direction on the face)
    void apply_Hv_free(const DVec& v, const unsigned char* free_var, DVec& res) {
        res.resize(v.size());
        m_ops->scale(1.0, v, res, free_var);
        std::vector<double> alpha(m_m, 0.0), ysf(m_m, 0.0);
        std::vector<char> use(m_m, 0);
        double theta = 1.0;
        bool have_theta = false;
        int j = m_ptr % m_m;
        for (int i = 0; i < m_ncorr; i++) {
            j = (j + m_m - 1) % m_m;
            const double ys = m_ops->dot((*m_s)[j], (*m_y)[j], free_var), yy = m_ops->dot((*m_y)[j], (*m_y)[j], free_var);
            ysf[j] = ys;
            use[j] = ys > std::numeric_limits<double>::epsilon() * yy;
            if (!use[j]) continue;
            if (!have_theta) { theta = yy / ys; have_theta = true; }
            alpha[j] = m_ops->dot((*m_s)[j], res, free_var) / ys;
            m_ops->axpy(-alpha[j], (*m_y)[j], res, free_var);
        }
        m_ops->scale(1.0 / theta, res, res);
        for (int i = 0; i < m_ncorr; i++) {
            if (use[j]) {
                const double b = m_ops->dot((*m_y)[j], res, free_var) / ysf[j];
                m_ops->axpy(alpha[j] - b, (*m_s)[j], res, free_var);
            }
            j = (j + 1) % m_m;
        }
    }
    int get_m() { return m_m; }
    int num_corrections() const { return m_ncorr; }
    double theta() const { return m_theta; }
private:
    int m_m;
    double m_theta;
    std::shared_ptr<std::vector<DVec>> m_s, m_y;
    Vector m_ys, m_alpha;
    int m_ncorr, m_ptr;
    DevOps* m_ops;
};

class DevLBFGSBSolver {
public:
    DevLBFGSBSolver(const LBFGSBParam& param, DevOps* ops) : m_param(param), m_ops(ops), m_free(nullptr), m_nfree(0) { m_param.check_param(); }
    ~DevLBFGSBSolver() { moihgp_dvec_free(m_free); }
    DevLBFGSBSolver(const DevLBFGSBSolver&) = delete;
    DevLBFGSBSolver& operator=(const DevLBFGSBSolver&) = delete;

    // Minimise f over lb <= x <= ub; `double f(const DVec& x, DVec& grad)`.  Same flow as LBFGSBSolver::minimize (lbfgsb.hpp).
    template <typename Foo>
    int minimize(Foo& f, DVec& x, double& fx, const DVec& lb, const DVec& ub) {
        const size_t n = x.size();
        if (lb.size() != n || ub.size() != n) throw std::invalid_argument("'lb' and 'ub' must have the same size as 'x'");
        moihgp_dvec_ctx* c = m_ops->ctx();
        dv_check(moihgp_dvec_clamp(c, n, x.data(), lb.data(), ub.data()), "dvec_clamp");
        {
            PhaseTimer pt("solver: storage");
            m_bfgs.reset(m_ops, n, m_param.m);
            for (DVec* v : {&m_grad, &m_xp, &m_gradp, &m_drt, &m_xt, &m_s, &m_y}) v->resize(n);
        }
        if (m_nfree != n) { moihgp_dvec_free(m_free); m_free = moihgp_dvec_alloc_mask(n); if (!m_free) throw std::runtime_error("moihgp_dvec_alloc_mask failed"); m_nfree = n; }
        std::vector<double> fpast(m_param.past > 0 ? m_param.past : 0);
        m_ops->sync();
        fx = f(x, m_grad);
        auto pgn = [&](const DVec& xx, const DVec& gg) { double r; dv_check(moihgp_dvec_proj_grad_norm(c, n, xx.data(), gg.data(), lb.data(), ub.data(), &r), "proj_grad_norm"); return r; };
        double pg = pgn(x, m_grad);
        if (m_param.past > 0) fpast[0] = fx;
        if (pg <= m_param.epsilon || pg <= m_param.epsilon_rel * std::sqrt(m_ops->dot(x, x))) return 1;
        const double eps = std::numeric_limits<double>::epsilon();
        int k = 1;
        for (;;) {
            double slope;
            {
                PhaseTimer pt("solver: direction (two-loop on the face)");
                m_ops->copy(x, m_xp); m_ops->copy(m_grad, m_gradp);
                dv_check(moihgp_dvec_active_set(c, n, x.data(), m_grad.data(), lb.data(), ub.data(), m_free), "active_set");
                m_bfgs.apply_Hv_free(m_grad, m_free, m_drt);
                m_ops->scale(-1.0, m_drt, m_drt);
                slope = m_ops->dot(m_grad, m_drt);
            }
            if (!(slope < 0.0)) {                                      // not a descent direction: steepest descent on the face
                m_ops->scale(-1.0, m_grad, m_drt, m_free);
                slope = m_ops->dot(m_grad, m_drt);
            }
            const double dn = std::sqrt(m_ops->dot(m_drt, m_drt));
            if (dn == 0.0) return k;
            double step = std::min(1.0, m_param.max_step / dn);
            if (k == 1 && m_bfgs.num_corrections() == 0) step = std::min(step, 1.0 / dn);
            bool ok = false;
            double ft = fx;
            for (int ls = 0; ls < m_param.max_linesearch; ls++) {
                double dec = 0.0;
                dv_check(moihgp_dvec_proj_step(c, n, m_xp.data(), m_drt.data(), step, lb.data(), ub.data(), m_gradp.data(), m_xt.data(), &dec), "proj_step");
                ft = f(m_xt, m_grad);
                if (std::isfinite(ft) && ft <= fx + m_param.ftol * dec) { ok = true; break; }
                step *= 0.5;
                if (step < m_param.min_step) break;
            }
            if (!ok) { fx = f(m_xp, m_grad); m_ops->copy(m_xp, x); m_ops->sync(); return k; }
            m_ops->copy(m_xt, x); fx = ft;
            pg = pgn(x, m_grad);
            if (pg <= m_param.epsilon || pg <= m_param.epsilon_rel * std::sqrt(m_ops->dot(x, x))) return k;
            if (m_param.past > 0) {
                const double fxd = fpast[k % m_param.past];
                if (k >= m_param.past && std::fabs(fxd - fx) <= m_param.delta * std::max(std::max(std::fabs(fx), std::fabs(fxd)), 1.0)) return k;
                fpast[k % m_param.past] = fx;
            }
            if (m_param.max_iterations != 0 && k >= m_param.max_iterations) return k;
            {
                PhaseTimer pt("solver: correction pair");
                m_ops->sub(x, m_xp, m_s); m_ops->sub(m_grad, m_gradp, m_y);
                if (m_ops->dot(m_s, m_y) > eps * m_ops->dot(m_y, m_y)) m_bfgs.add_correction(m_s, m_y);
            }
            k++;
        }
    }
    DevBFGSMat getBFGSMat() { return m_bfgs; }
private:
    LBFGSBParam m_param;
    DevOps* m_ops;
    DevBFGSMat m_bfgs;
    DVec m_grad, m_xp, m_gradp, m_drt, m_xt, m_s, m_y;
    unsigned char* m_free;
    size_t m_nfree;
};

}  // namespace opt

// OnlineObjective of moihgp_online.hpp (moihgp_online.h:18-116) with params / grad on the device: update and window evaluation through
// moihgp_update_dev / moihgp_window_eval_dev, the proximal term through the device BFGS matrix.  The carried window-start state and the
// window itself stay on the host side of the ABI (they are small: L d, L P d, W M doubles).
template <typename StateSpace>
class OnlineObjectiveDev {
public:
    typedef std::vector<double> Vector;
    OnlineObjectiveDev(MOIHGP<StateSpace>* gp, opt::DevOps* ops, const double& gamma, const size_t& windowsize) : _gp(gp), _ops(ops) {
        _dim = gp->getIGPDim(); _num_param = gp->getNumParam(); _igp_num_param = gp->getNumIGPParam();
        _num_output = gp->getNumOutput(); _num_latent = gp->getNumLatent();
        _gamma = gamma; _windowsize = windowsize;
        oldparams.resize(_num_param);
        opt::dv_check(moihgp_get_params_dev(gp->handle(), oldparams.data()), "get_params_dev");
        _x.assign(_num_latent * _dim, 0.0); _dx.assign(_num_latent * _igp_num_param * _dim, 0.0);
        _xd.resize(_x.size()); _dxd.resize(_dx.size()); _loss.resize(1); _dparams.resize(_num_param); _g.resize(_num_param);
        ma.assign(_num_output, 0.0);
    }
    // moihgp_online.h:40-72
    double operator()(const opt::DVec& params, opt::DVec& grad) {
        evaluations++;
        _ops->sub(params, oldparams, _dparams);
        // (no host synchronisation: the handle's `_on` entries take their place behind this context's stream by an event)
        // :43 `_gp->update(params)` -- skipped when the handle already holds exactly these parameters: a solve's first evaluation is at the point
        // the previous solve's last one was (the optimiser hands its final iterate on), and update() is a function of the parameters alone
        // (33 ms of a 206 ms learner tick at M = L = 4096: one polar factor in six).  Two vector passes to find out.
        bool same = false;
        if (_have_last) {
            opt::PhaseTimer pt("objective: same parameters as the last update?");
            _ops->sub(params, _last, _g);
            same = _ops->dot(_g, _g) == 0.0;
        }
        if (!same) {
            _have_last = false;                                                      // (a failing update leaves the handle half-way: the next evaluation must not skip it)
            { opt::PhaseTimer pt("objective: update_dev"); opt::dv_check(moihgp_update_dev_on(_gp->handle(), params.data(), _ops->stream()), "update_dev"); }
            if (opt::phases().enabled) { opt::phases().seconds["objective: Newton-Schulz steps of the polar factor (count, not ms)"] += 1e-3 * moihgp_polar_iterations(_gp->handle()); opt::phases().calls["objective: Newton-Schulz steps of the polar factor (count, not ms)"]++; }
            _last.resize(_num_param);
            _ops->copy(params, _last);
            _have_last = true;
        } else updates_skipped++;

        double loss;
        {
            opt::PhaseTimer pt("objective: proximal term (B dp)");
            if (bfgs_mat.get_m() > 0) bfgs_mat.apply_Hv(_dparams, _gamma, grad);     // :45-48: grad = Bp
            else { grad.resize(_num_param); _ops->scale(1.0, _dparams, grad); }      // :51
            loss = 0.5 * _ops->dot(_dparams, grad);                                  // :53
        }
        if (!Y.empty()) {
            if (_window_dirty) {
                opt::PhaseTimer pt("objective: window_set");
                _Yflat.resize(Y.size() * _num_output);
                size_t t = 0;
                for (std::list<Vector>::iterator it = Y.begin(); it != Y.end(); ++it, ++t)
                    for (size_t m = 0; m < _num_output; m++) _Yflat[t * _num_output + m] = (*it)[m] - ma[m];       // :63
                const int rc = moihgp_window_set(_gp->handle(), _Yflat.data(), Y.size());
                if (rc != 0 && rc != 3) throw std::runtime_error(std::string("moihgp_window_set: ") + moihgp_last_error());
                _per_tick = rc == 3;      // missing outputs beyond the batched kernel's limits: the reference's loop, tick by tick (as the host learner)
                _ops->upload(_x, _xd); _ops->upload(_dx, _dxd);
                _window_dirty = false;
            }
            if (_per_tick) {
                opt::PhaseTimer pt("objective: window, tick by tick (host)");
                Vector gh(_num_param);
                const double wl = window_loop_per_tick(_gp->handle(), _Yflat.data(), Y.size(), _num_output, _x, _dx, gh);
                _ops->upload(gh, _g);
                loss += wl;
            } else {
                opt::PhaseTimer pt("objective: window_eval_dev");
                // ordered behind this context's stream on entry, in front of it on return: the download and the axpy below follow it there
                opt::dv_check(moihgp_window_eval_dev_on(_gp->handle(), _xd.data(), _dxd.data(), _loss.data(), _g.data(), nullptr, nullptr, _ops->stream()), "window_eval_dev");   // :61-70
                std::vector<double> l1;
                _ops->download(_loss, l1);
                loss += l1[0];
            }
            _ops->axpy(1.0, _g, grad);
        }
        return loss;
    }
    // moihgp_online.h:75-93 (as OnlineObjective::push_back)
    void push_back(const Vector& y) {
        Y.push_back(y);
        ma.assign(_num_output, 0.0);
        for (std::list<Vector>::iterator it = Y.begin(); it != Y.end(); ++it)
            for (size_t m = 0; m < _num_output; m++) ma[m] += (*it)[m];
        for (size_t m = 0; m < _num_output; m++) ma[m] /= double(Y.size());
        while (Y.size() > _windowsize) {
            Y.pop_front();
            Vector yc(_num_output), xnew(_x.size()), dxnew(_dx.size());
            for (size_t m = 0; m < _num_output; m++) yc[m] = Y.front()[m] - ma[m];
            gp32_step2(_gp->handle(), _x.data(), yc.data(), _dx.data(), xnew.data(), dxnew.data());   // :89
            _x = xnew; _dx = dxnew;
        }
        _window_dirty = true;
    }
    opt::DVec oldparams;
    opt::DevBFGSMat bfgs_mat;
    std::list<Vector> Y;
    Vector ma;
    size_t evaluations = 0;          // objective evaluations so far (diagnosis / bench)
    size_t updates_skipped = 0;      // ... of which found the handle up to date
    void invalidate_update_cache() { _have_last = false; }      // somebody else has called update() on the handle
private:
    MOIHGP<StateSpace>* _gp;
    opt::DevOps* _ops;
    size_t _num_output, _num_latent, _igp_num_param, _num_param, _dim, _windowsize;
    double _gamma;
    Vector _x, _dx, _Yflat;
    opt::DVec _xd, _dxd, _loss, _dparams, _g, _last;
    bool _window_dirty = true, _have_last = false, _per_tick = false;
};

// MOIHGPOnlineLearning of moihgp_online.hpp (moihgp_online.h:118-255) whose parameter vector, gradient and optimiser state live on the device.
template <typename StateSpace>
class MOIHGPOnlineLearningDev {
public:
    typedef std::vector<double> Vector;
    MOIHGPOnlineLearningDev(const double& dt, const size_t& num_output, const size_t& num_latent, const double& gamma, const size_t& windowsize, const bool& threading) {
        _num_output = num_output; _num_latent = num_latent;
        _moihgp = new MOIHGP<StateSpace>(dt, num_output, num_latent, threading);
        _dim = _moihgp->getIGPDim(); _igp_num_param = _moihgp->getNumIGPParam(); _num_param = _moihgp->getNumParam();
        Vector lb(_num_param), ub(_num_param);
        const size_t nu = num_output * num_latent;
        for (size_t i = 0; i < nu; i++) { lb[i] = -1e+4; ub[i] = 1e+4; }                                  // moihgp_online.h:135-136
        for (size_t i = nu; i < nu + num_latent; i++) { lb[i] = 1e-4; ub[i] = 1e+4; }                     // :137-138
        for (size_t i = nu + num_latent; i < _num_param; i++) { lb[i] = 1e-4; ub[i] = 1e+2; }             // :139-140
        _ops.upload(lb, _lb); _ops.upload(ub, _ub);
        x.assign(num_latent * _dim, 0.0);
        dx.assign(num_latent * _igp_num_param * _dim, 0.0);
        _windowsize = windowsize < 1 ? 1 : windowsize;
        _params.resize(_num_param);
        opt::dv_check(moihgp_get_params_dev(_moihgp->handle(), _params.data()), "get_params_dev");
        _LBFGSB_param.m = 10; _LBFGSB_param.max_iterations = 5; _LBFGSB_param.max_linesearch = 20; _LBFGSB_param.max_step = 1e-1;     // :153-159
        _LBFGSB_param.ftol = 1e-8; _LBFGSB_param.epsilon = 1e-8; _LBFGSB_param.epsilon_rel = 1e-8;
        moihgp_set_option(_moihgp->handle(), "polar_warm_start", 1);        // consecutive evaluations share their outlying subspace: one pass fewer per update
        _solver = new opt::DevLBFGSBSolver(_LBFGSB_param, &_ops);
        _obj = new OnlineObjectiveDev<StateSpace>(_moihgp, &_ops, gamma, _windowsize);
    }
    ~MOIHGPOnlineLearningDev() { delete _obj; delete _solver; delete _moihgp; }
    MOIHGPOnlineLearningDev(const MOIHGPOnlineLearningDev&) = delete;
    MOIHGPOnlineLearningDev& operator=(const MOIHGPOnlineLearningDev&) = delete;

    // moihgp_online.h:173-187
    Vector step(const Vector& y) {
        Vector yhat(_num_output), yc(_num_output), xnew(x.size());
        { opt::PhaseTimer pt("step: push_back (window, carried state)"); _obj->push_back(y); }
        {
            opt::PhaseTimer pt("step: filter the new observation");
            for (size_t m = 0; m < _num_output; m++) yc[m] = y[m] - _obj->ma[m];
            gp32_step3(_moihgp->handle(), x.data(), yc.data(), xnew.data(), yhat.data());               // :178
            for (size_t m = 0; m < _num_output; m++) yhat[m] += _obj->ma[m];                            // :179
        }
        x = xnew;
        dx.assign(dx.size(), 0.0);                                                                  // :181
        _obj->bfgs_mat = _solver->getBFGSMat();                                                     // :182 (shares the pairs: no copy)
        _ops.copy(_params, _obj->oldparams);                                                        // :183
        double fx;
        last_iterations = _solver->minimize(*_obj, _params, fx, _lb, _ub);                          // :185
        last_fx = fx;
        return yhat;
    }
    Vector getParams() { return _moihgp->getParams(); }
    size_t getNumParam() { return _num_param; }
    // (not in the reference: start from given parameters instead of the constructor's random draw -- tests, warm starts)
    void setParams(const Vector& p) {
        _moihgp->update(p);
        _obj->invalidate_update_cache();
        opt::dv_check(moihgp_get_params_dev(_moihgp->handle(), _params.data()), "get_params_dev");
        _ops.copy(_params, _obj->oldparams);
        _ops.sync();
    }
    OnlineObjectiveDev<StateSpace>& objective() { return *_obj; }
    opt::DevOps& ops() { return _ops; }
    opt::DVec& params_dev() { return _params; }
    Vector x, dx;
    int last_iterations = 0;
    double last_fx = 0.0;
private:
    opt::DevOps _ops;
    MOIHGP<StateSpace>* _moihgp;
    size_t _dim, _num_output, _num_latent, _num_param, _igp_num_param, _windowsize;
    opt::DVec _params, _lb, _ub;
    opt::LBFGSBParam _LBFGSB_param;
    opt::DevLBFGSBSolver* _solver;
    OnlineObjectiveDev<StateSpace>* _obj;
};

}  // namespace moihgp

#endif
