// moihgp_online.hpp -- Eigen-free C++ online learner over libmoihgp.so, mirroring the reference's
// moihgp::OnlineObjective<SS> / moihgp::MOIHGPOnlineLearning<SS> (reference moihgp/include/moihgp/moihgp_online.h:18-255):
// same constructor arguments, same members (`x`, `dx`, `Y`, `ma`, `oldparams`, `bfgs_mat`), same window / carry semantics,
// same proximal term, same bounds and solver settings.  Differences, all on the host side:
//   * containers are std::vector<double> (flat [L][d] / [L][P][d] states), not Eigen types;
//   * the objective's loop over the window (moihgp_online.h:61-70) is ONE device call (moihgp_window_set / moihgp_window_eval), windows
//     with missing outputs included; only a window beyond that call's limits (rc 3) runs the loop through the per-tick ABI;
//   * the optimiser is this repo's projected L-BFGS (lbfgsb.hpp), not LBFGS++: same problem, different iterates.
// To run the reference's own learner header (with LBFGS++ and Eigen) on top of this library instead, see
// include/moihgp_cxx/compat/ and INTEGRATION.md.
#ifndef MOIHGP_CXX_MOIHGP_ONLINE_HPP_
#define MOIHGP_CXX_MOIHGP_ONLINE_HPP_

#include <list>
#include <stdexcept>
#include <string>
#include <vector>

#include "lbfgsb.hpp"
#include "moihgp.hpp"

namespace moihgp {

template <typename StateSpace>
class OnlineObjective {
public:
    typedef std::vector<double> Vector;

    OnlineObjective(MOIHGP<StateSpace>* gp, const double& gamma, const size_t& windowsize) {        // moihgp_online.h:23-37
        _gp = gp;
        _dim = _gp->getIGPDim();
        _num_param = _gp->getNumParam();
        _igp_num_param = _gp->getNumIGPParam();
        _num_output = _gp->getNumOutput();
        _num_latent = _gp->getNumLatent();
        oldparams = _gp->getParams();
        _gamma = gamma;
        _windowsize = windowsize;
        _x.assign(_num_latent * _dim, 0.0);
        _dx.assign(_num_latent * _igp_num_param * _dim, 0.0);
        ma.assign(_num_output, 0.0);
    }

    // moihgp_online.h:40-72: loss = 1/2 dp' Bp + sum over the window of negLogLikelihood at the pre-step state
    double operator()(const Vector& params, Vector& grad) {
        Vector dparams(_num_param), Bp;
        for (size_t i = 0; i < _num_param; i++) dparams[i] = params[i] - oldparams[i];
        if (!_have_last || params != _last) {                                    // :43 (skipped when the handle already holds exactly these parameters:
            _have_last = false;                                                  //      a solve's first evaluation is at the previous solve's last point, and
            _gp->update(params);                                                 //      update() depends on the parameters alone; a throwing update
            _last = params; _have_last = true;                                   //      leaves the cache empty)
        }
        if (bfgs_mat.get_m() > 0) bfgs_mat.apply_Hv(dparams, _gamma, Bp);        // :45-48
        else Bp = dparams;                                                       // :51
        double loss = 0.5 * opt::dot(dparams, Bp);                               // :53
        grad = Bp;                                                               // :54
        if (!Y.empty()) {
            if (_window_dirty) {                                                 // upload (y_t - ma) once per window
                _Yflat.resize(Y.size() * _num_output);
                size_t t = 0;
                for (std::list<Vector>::iterator it = Y.begin(); it != Y.end(); ++it, ++t)
                    for (size_t m = 0; m < _num_output; m++) _Yflat[t * _num_output + m] = (*it)[m] - ma[m];   // :63
                const int rc = moihgp_window_set(_gp->handle(), _Yflat.data(), Y.size());
                if (rc != 0 && rc != 3) throw std::runtime_error(std::string("moihgp_window_set: ") + moihgp_last_error());
                _per_tick = rc == 3;      // missing outputs beyond the batched kernel's limits: the reference's loop, tick by tick
                _window_dirty = false;
            }
            double wloss = 0.0;
            _g.resize(_num_param);
            if (_per_tick) wloss = window_loop_per_tick(_gp->handle(), _Yflat.data(), Y.size(), _num_output, _x, _dx, _g);
            else if (moihgp_window_eval(_gp->handle(), _x.data(), _dx.data(), &wloss, _g.data(), nullptr, nullptr) != 0)   // :61-70
                throw std::runtime_error(std::string("moihgp_window_eval: ") + moihgp_last_error());
            loss += wloss;
            for (size_t i = 0; i < _num_param; i++) grad[i] += _g[i];
        }
        return loss;
    }

    // moihgp_online.h:75-93: append, recompute the window mean, and while the window is too long drop its oldest tick and
    // advance the carried start state by one step on the NEW front (minus the mean), as the reference does
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
            _x = xnew;
            _dx = dxnew;
        }
        _window_dirty = true;
    }

    Vector oldparams;
    opt::BFGSMat bfgs_mat;
    std::list<Vector> Y;
    Vector ma;

private:
    size_t _num_output, _num_latent, _igp_num_param, _num_param, _dim, _windowsize;
    double _gamma;
    MOIHGP<StateSpace>* _gp;
    Vector _x, _dx, _Yflat, _g, _last;
    bool _window_dirty = true, _per_tick = false, _have_last = false;

public:
    void invalidate_update_cache() { _have_last = false; }      // somebody else has called update() on the handle
};

template <typename StateSpace>
class MOIHGPOnlineLearning {
public:
    typedef std::vector<double> Vector;

    MOIHGPOnlineLearning(const double& dt, const size_t& num_output, const size_t& num_latent, const double& gamma,
                         const size_t& windowsize, const bool& threading) {                          // moihgp_online.h:123-162
        _dt = dt; _num_output = num_output; _num_latent = num_latent; _threading = threading;
        _moihgp = new MOIHGP<StateSpace>(dt, num_output, num_latent, threading);
        _dim = _moihgp->getIGPDim();
        _igp_num_param = _moihgp->getNumIGPParam();
        _num_param = _moihgp->getNumParam();
        _lb.assign(_num_param, 0.0); _ub.assign(_num_param, 0.0);
        const size_t nu = _num_output * _num_latent;
        for (size_t i = 0; i < nu; i++) { _lb[i] = -1e+4; _ub[i] = 1e+4; }                            // :135-136
        for (size_t i = nu; i < nu + _num_latent; i++) { _lb[i] = 1e-4; _ub[i] = 1e+4; }              // :137-138
        for (size_t i = nu + _num_latent; i < _num_param; i++) { _lb[i] = 1e-4; _ub[i] = 1e+2; }      // :139-140 (sigma and the latents)
        x.assign(_num_latent * _dim, 0.0);
        dx.assign(_num_latent * _igp_num_param * _dim, 0.0);
        _gamma = gamma;
        _windowsize = windowsize < 1 ? 1 : windowsize;                                              // :144-151
        _params = _moihgp->getParams();
        _LBFGSB_param.m = 10;                                                                       // :153-159
        _LBFGSB_param.max_iterations = 5;
        _LBFGSB_param.max_linesearch = 20;
        _LBFGSB_param.max_step = 1e-1;
        _LBFGSB_param.ftol = 1e-8;
        _LBFGSB_param.epsilon = 1e-8;
        _LBFGSB_param.epsilon_rel = 1e-8;
        _solver = new opt::LBFGSBSolver(_LBFGSB_param);
        _obj = new OnlineObjective<StateSpace>(_moihgp, _gamma, _windowsize);
    }
    ~MOIHGPOnlineLearning() { delete _obj; delete _solver; delete _moihgp; }
    MOIHGPOnlineLearning(const MOIHGPOnlineLearning&) = delete;
    MOIHGPOnlineLearning& operator=(const MOIHGPOnlineLearning&) = delete;

    // moihgp_online.h:173-187: filter the new observation with the current parameters, then re-fit on the window
    Vector step(const Vector& y) {
        Vector yhat(_num_output), yc(_num_output), xnew(x.size());
        _obj->push_back(y);
        for (size_t m = 0; m < _num_output; m++) yc[m] = y[m] - _obj->ma[m];
        gp32_step3(_moihgp->handle(), x.data(), yc.data(), xnew.data(), yhat.data());               // :178
        for (size_t m = 0; m < _num_output; m++) yhat[m] += _obj->ma[m];                            // :179
        x = xnew;
        dx.assign(dx.size(), 0.0);                                                                  // :181 (`dx = dxnew`, zeros there)
        _obj->bfgs_mat = _solver->getBFGSMat();                                                     // :182
        _obj->oldparams = _params;                                                                  // :183
        double fx;
        _solver->minimize(*_obj, _params, fx, _lb, _ub);                                            // :185
        return yhat;
    }

    Vector getParams() { return _moihgp->getParams(); }
    // (not in the reference: start from given parameters instead of the constructor's random draw -- tests, warm starts)
    void setParams(const Vector& p) { _moihgp->update(p); _obj->invalidate_update_cache(); _params = _moihgp->getParams(); _obj->oldparams = _params; }
    size_t getNumParam() { return _num_param; }
    size_t getNumOutput() { return _num_output; }
    size_t getNumLatent() { return _num_latent; }
    size_t getNumIGPParam() { return _igp_num_param; }
    size_t getIGPDim() { return _dim; }
    size_t getWindowsize() { return _windowsize; }
    OnlineObjective<StateSpace>& objective() { return *_obj; }     // (not in the reference: lets tests evaluate the objective)

    Vector x;      // [L][d]
    Vector dx;     // [L][P][d]

private:
    MOIHGP<StateSpace>* _moihgp;
    bool _threading;
    double _dt, _gamma;
    size_t _dim, _num_output, _num_latent, _num_param, _igp_num_param, _windowsize;
    Vector _params, _lb, _ub;
    opt::LBFGSBParam _LBFGSB_param;
    opt::LBFGSBSolver* _solver;
    OnlineObjective<StateSpace>* _obj;
};

}  // namespace moihgp

#endif
