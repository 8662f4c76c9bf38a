// moihgp.hpp -- C++ host surface `moihgp::MOIHGP<StateSpace>` over the C ABI of libmoihgp.so.
//
// Mirrors the public interface of the reference class template (reference
// moihgp/include/moihgp/moihgp.h:76-757): same constructor arguments, the four step() overloads, the two
// negLogLikelihood() overloads, update(), getParams() and the size getters, with the same argument meaning.
// The reference spells its containers with Eigen (std::vector<Eigen::VectorXd>, Eigen::VectorXd); this
// header is generic over any vector type with size(), resize(n) and operator[] -- Eigen::VectorXd and
// std::vector<double> both qualify -- so it needs no Eigen itself.  All arithmetic runs in HIP kernels
// behind the C ABI (include/moihgp.h); there is no host fallback: construction throws std::runtime_error
// when no GPU is usable.
#ifndef MOIHGP_CXX_MOIHGP_HPP_
#define MOIHGP_CXX_MOIHGP_HPP_

#include <cstddef>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

extern "C" {
#include "../moihgp.h"
}

namespace moihgp {

// Tags standing in for the reference's StateSpace template arguments (matern32ss.h:13, matern52ss.h:13).
struct Matern32StateSpace { static constexpr int kernel_id = MOIHGP_MATERN32; };
struct Matern52StateSpace { static constexpr int kernel_id = MOIHGP_MATERN52; };
// Stacked state: the sum of J (2, 3 or 4) Matern components behind one output, state dim 2J / 3J, 2J + 1 hyper-parameters per latent
// (BASELINE.json's d = 6 / d = 12 shapes; not a model the reference ships -- what its IHGP<StateSpace> template computes for a
// block-diagonal StateSpace built from its own component models, include/moihgp.h MOIHGP_STACK).
template <int J> struct StackedMatern32StateSpace { static constexpr int kernel_id = MOIHGP_STACK(MOIHGP_MATERN32, J); };
template <int J> struct StackedMatern52StateSpace { static constexpr int kernel_id = MOIHGP_STACK(MOIHGP_MATERN52, J); };

template <typename StateSpace>
class MOIHGP {
public:
    // moihgp.h:81  MOIHGP(dt, num_output, num_latent, threading)
    MOIHGP(const double& dt, const size_t& num_output, const size_t& num_latent, const bool& threading = false)
        : _num_output(num_output), _num_latent(num_latent) {
        _gp = moihgp_new(StateSpace::kernel_id, dt, num_output, num_latent);
        if (!_gp) throw std::runtime_error(std::string("moihgp::MOIHGP: ") + moihgp_last_error());
        // no pthread fan-out here (moihgp.h:184-214), but the flag decides what negLogLikelihood(x, y, dx, grad) RETURNS in the
        // reference (moihgp.h:590 vs :597-607) and the library honours that; it applies the L < 2 override of :128-135 itself
        moihgp_set_threading(_gp, threading ? 1 : 0);
        _dim = gp32_igp_dim(_gp);
        _igp_num_param = gp32_num_igp_param(_gp);
        _num_param = gp32_num_param(_gp);
        _x.resize(_num_latent * _dim); _xnew.resize(_num_latent * _dim);
        _dx.resize(_num_latent * _igp_num_param * _dim); _dxnew.resize(_dx.size());
        _y.resize(_num_output); _yhat.resize(_num_output);
        _p.resize(_num_param);
    }
    ~MOIHGP() { if (_gp) moihgp_del(_gp); }
    MOIHGP(const MOIHGP&) = delete;
    MOIHGP& operator=(const MOIHGP&) = delete;

    // (y may be an expression such as `y - mean` of the caller's vector library, so its type is independent of yhat's / grad's)
    // moihgp.h:148  step(x, y, dx, xnew, yhat, dxnew)
    template <class VecList, class VecIn, class Vec, class VecListList>
    void step(const VecList& x, const VecIn& y, const VecListList& dx, VecList& xnew, Vec& yhat, VecListList& dxnew) {
        pack_x(x); pack_y(y); pack_dx(dx);
        gp32_step1(_gp, _x.data(), _y.data(), _dx.data(), _xnew.data(), _yhat.data(), _dxnew.data());
        unpack_x(xnew); unpack_y(yhat); unpack_dx(dxnew);
    }
    // moihgp.h:229  step(x, y, dx, xnew, dxnew)
    template <class VecList, class Vec, class VecListList>
    void step(const VecList& x, const Vec& y, const VecListList& dx, VecList& xnew, VecListList& dxnew) {
        pack_x(x); pack_y(y); pack_dx(dx);
        gp32_step2(_gp, _x.data(), _y.data(), _dx.data(), _xnew.data(), _dxnew.data());
        unpack_x(xnew); unpack_dx(dxnew);
    }
    // moihgp.h:304  step(x, y, xnew, yhat)
    template <class VecList, class VecIn, class Vec>
    void step(const VecList& x, const VecIn& y, VecList& xnew, Vec& yhat) {
        pack_x(x); pack_y(y);
        gp32_step3(_gp, _x.data(), _y.data(), _xnew.data(), _yhat.data());
        unpack_x(xnew); unpack_y(yhat);
    }
    // moihgp.h:381  step(x, xnew, yhat)   (prediction only)
    template <class VecList, class Vec>
    void step(VecList& x, VecList& xnew, Vec& yhat) {
        pack_x(x);
        gp32_step4(_gp, _x.data(), _xnew.data(), _yhat.data());
        unpack_x(xnew); unpack_y(yhat);
    }
    // moihgp.h:431  update(params)
    template <class Vec>
    void update(const Vec& params) {
        for (size_t i = 0; i < _num_param; i++) _p[i] = params[i];
        gp32_update(_gp, _p.data());
    }
    // moihgp.h:460  negLogLikelihood(x, y, dx, grad)
    template <class VecList, class VecIn, class Vec, class VecListList>
    double negLogLikelihood(const VecList& x, const VecIn& y, const VecListList& dx, Vec& grad) {
        pack_x(x); pack_y(y); pack_dx(dx);
        double loss = gp32_lik1(_gp, _x.data(), _y.data(), _dx.data(), _p.data());
        if ((size_t)grad.size() != _num_param) grad.resize(_num_param);
        for (size_t i = 0; i < _num_param; i++) grad[i] = _p[i];
        return loss;
    }
    // moihgp.h:614  negLogLikelihood(x, y)
    template <class VecList, class Vec>
    double negLogLikelihood(VecList& x, const Vec& y) {
        pack_x(x); pack_y(y);
        return gp32_lik2(_gp, _x.data(), _y.data());
    }
    // moihgp.h:721  getParams()  -> [U row-major | S | sigma | (magnitude, lengthscale, noise) x L]
    // The reference returns an Eigen::VectorXd; this returns a value that converts to whatever vector type the caller
    // initialises or assigns from it (std::vector<double>, Eigen::VectorXd, ...: anything with resize(n) and operator[]).
    struct Params {
        std::vector<double> values;
        operator const std::vector<double>&() const { return values; }
        template <class V, class = decltype(std::declval<V&>().resize(std::declval<size_t>())), class = decltype(std::declval<V&>()[0])>
        operator V() const {
            V v;
            v.resize(values.size());
            for (size_t i = 0; i < values.size(); i++) v[i] = values[i];
            return v;
        }
        size_t size() const { return values.size(); }
        double operator[](size_t i) const { return values[i]; }
        std::vector<double>::const_iterator begin() const { return values.begin(); }
        std::vector<double>::const_iterator end() const { return values.end(); }
    };
    Params getParams() {
        Params p;
        p.values.resize(_num_param);
        gp32_get_params(_gp, p.values.data());
        return p;
    }
    size_t getIGPDim() { return _dim; }                   // moihgp.h:691
    size_t getNumOutput() { return _num_output; }         // moihgp.h:697
    size_t getNumLatent() { return _num_latent; }         // moihgp.h:703
    size_t getNumParam() { return _num_param; }           // moihgp.h:709
    size_t getNumIGPParam() { return _igp_num_param; }    // moihgp.h:715
    moihgp_gp* handle() { return _gp; }                   // for the batched entry points of include/moihgp.h

private:
    template <class VecList> void pack_x(const VecList& x) {
        for (size_t l = 0; l < _num_latent; l++) for (size_t i = 0; i < _dim; i++) _x[l * _dim + i] = x[l][i];   // wrapper.cpp:59-64
    }
    template <class VecListList> void pack_dx(const VecListList& dx) {
        for (size_t l = 0; l < _num_latent; l++) for (size_t p = 0; p < _igp_num_param; p++) for (size_t i = 0; i < _dim; i++)
            _dx[(l * _igp_num_param + p) * _dim + i] = dx[l][p][i];                                               // wrapper.cpp:65-71
    }
    template <class Vec> void pack_y(const Vec& y) { for (size_t m = 0; m < _num_output; m++) _y[m] = y[m]; }
    template <class VecList> void unpack_x(VecList& xnew) {
        if ((size_t)xnew.size() != _num_latent) xnew.resize(_num_latent);
        for (size_t l = 0; l < _num_latent; l++) {
            if ((size_t)xnew[l].size() != _dim) xnew[l].resize(_dim);
            for (size_t i = 0; i < _dim; i++) xnew[l][i] = _xnew[l * _dim + i];
        }
    }
    template <class VecListList> void unpack_dx(VecListList& dxnew) {
        if ((size_t)dxnew.size() != _num_latent) dxnew.resize(_num_latent);
        for (size_t l = 0; l < _num_latent; l++) {
            if ((size_t)dxnew[l].size() != _igp_num_param) dxnew[l].resize(_igp_num_param);
            for (size_t p = 0; p < _igp_num_param; p++) {
                if ((size_t)dxnew[l][p].size() != _dim) dxnew[l][p].resize(_dim);
                for (size_t i = 0; i < _dim; i++) dxnew[l][p][i] = _dxnew[(l * _igp_num_param + p) * _dim + i];
            }
        }
    }
    template <class Vec> void unpack_y(Vec& yhat) {
        if ((size_t)yhat.size() != _num_output) yhat.resize(_num_output);
        for (size_t m = 0; m < _num_output; m++) yhat[m] = _yhat[m];
    }

    moihgp_gp* _gp = nullptr;
    size_t _num_output, _num_latent, _dim = 0, _igp_num_param = 0, _num_param = 0;
    std::vector<double> _x, _xnew, _dx, _dxnew, _y, _yhat, _p;
};

// The learners' window loop (moihgp_online.h:61-70, moihgp_regression.h:42-50) through the per-tick reference ABI, for windows the
// batched objective refuses (moihgp_window_set rc 3: a tick with more missing outputs than its least-squares kernel takes):
//     for t < W:  loss += gpXX_lik1(x, y_t, dx, g);  grad += g;  (x, dx) <- gpXX_step2(x, y_t, dx)
// Y is [W][M] tick-major, x [L][d] and dx [L][P][d] the state before the window (left untouched); grad must hold num_param entries.
inline double window_loop_per_tick(moihgp_gp* gp, const double* Y, size_t W, size_t M, const std::vector<double>& x0, const std::vector<double>& dx0,
                                   std::vector<double>& grad) {
    std::vector<double> x(x0), dx(dx0), xn(x0.size()), dxn(dx0.size()), g(grad.size()), y(M);
    double loss = 0.0;
    for (size_t i = 0; i < grad.size(); i++) grad[i] = 0.0;
    for (size_t t = 0; t < W; t++) {
        for (size_t m = 0; m < M; m++) y[m] = Y[t * M + m];
        loss += gp32_lik1(gp, x.data(), y.data(), dx.data(), g.data());          // (the entry points dispatch on the handle's own model)
        for (size_t i = 0; i < grad.size(); i++) grad[i] += g[i];
        gp32_step2(gp, x.data(), y.data(), dx.data(), xn.data(), dxn.data());
        x.swap(xn); dx.swap(dxn);
    }
    return loss;
}

}  // namespace moihgp

#endif
