// moihgp_regression.hpp -- Eigen-free C++ batch learner over libmoihgp.so, mirroring the reference's
// moihgp::RegressionObjective<SS> / moihgp::MOIHGPRegression<SS> (reference moihgp/include/moihgp/moihgp_regression.h:17-202):
// fit(Y) minimises the summed negative log-likelihood of the whole series from a zero start state, predict(Y) filters it.
// Like the reference (moihgp_regression.h:34-52) the objective does NOT call update(params): literal, including the
// consequence that the model's parameters never change during fit().  Set `apply_params = true` to evaluate the objective at
// the parameters it is given.  Host-side differences from the reference: std::vector containers, one device call per objective
// evaluation (moihgp_window_*), this repo's optimiser (lbfgsb.hpp) instead of LBFGS++.
#ifndef MOIHGP_CXX_MOIHGP_REGRESSION_HPP_
#define MOIHGP_CXX_MOIHGP_REGRESSION_HPP_

#include <stdexcept>
#include <string>
#include <vector>

#include "lbfgsb.hpp"
#include "moihgp.hpp"

namespace moihgp {

template <typename StateSpace>
class RegressionObjective {
public:
    typedef std::vector<double> Vector;
    RegressionObjective(const size_t& num_data, MOIHGP<StateSpace>* gp) {                            // moihgp_regression.h:22-31
        _gp = gp;
        _dim = _gp->getIGPDim();
        _num_param = _gp->getNumParam();
        _igp_num_param = _gp->getNumIGPParam();
        _num_latent = _gp->getNumLatent();
        _num_output = _gp->getNumOutput();
        _num_data = num_data;
        Y.reserve(_num_data);
    }
    double operator()(const Vector& params, Vector& grad) {                                          // moihgp_regression.h:34-52
        if (apply_params) _gp->update(params);
        grad.assign(_num_param, 0.0);
        if (Y.empty()) return 0.0;
        if (_dirty) {
            _Yflat.resize(Y.size() * _num_output);
            for (size_t t = 0; t < Y.size(); t++) for (size_t m = 0; m < _num_output; m++) _Yflat[t * _num_output + m] = Y[t][m];
            const int rc = moihgp_window_set(_gp->handle(), _Yflat.data(), Y.size());
            if (rc != 0 && rc != 3) throw std::runtime_error(std::string("moihgp_window_set: ") + moihgp_last_error());
            _per_tick = rc == 3;          // missing outputs beyond the batched kernel's limits: the reference's loop, tick by tick
            _dirty = false;
        }
        Vector x(_num_latent * _dim, 0.0), dx(_num_latent * _igp_num_param * _dim, 0.0);            // :38-39 zero start
        if (_per_tick) return window_loop_per_tick(_gp->handle(), _Yflat.data(), Y.size(), _num_output, x, dx, grad);
        double loss = 0.0;
        if (moihgp_window_eval(_gp->handle(), x.data(), dx.data(), &loss, grad.data(), nullptr, nullptr) != 0)
            throw std::runtime_error(std::string("moihgp_window_eval: ") + moihgp_last_error());
        return loss;
    }
    void set_data(const std::vector<Vector>& data) { Y = data; _dirty = true; }
    std::vector<Vector> Y;
    bool apply_params = false;

private:
    size_t _dim, _num_param, _igp_num_param, _num_latent, _num_output, _num_data;
    MOIHGP<StateSpace>* _gp;
    Vector _Yflat;
    bool _dirty = true, _per_tick = false;
};

template <typename StateSpace>
class MOIHGPRegression {
public:
    typedef std::vector<double> Vector;
    MOIHGPRegression(const double& dt, const size_t& num_output, const size_t& num_latent, const size_t& num_data,
                     const bool& threading) {                                                      // moihgp_regression.h:80-108
        _dt = dt; _num_output = num_output; _num_latent = num_latent; _num_data = num_data; _threading = threading;
        _moihgp = new MOIHGP<StateSpace>(dt, num_output, num_latent, threading);
        _dim = _moihgp->getIGPDim();
        _num_param = _moihgp->getNumParam();
        _igp_num_param = _moihgp->getNumIGPParam();
        _lb.assign(_num_param, 0.0); _ub.assign(_num_param, 0.0);
        const size_t nu = _num_output * _num_latent;
        for (size_t i = 0; i < nu; i++) { _lb[i] = -1e+4; _ub[i] = 1e+4; }
        for (size_t i = nu; i < nu + _num_latent; i++) { _lb[i] = 1e-4; _ub[i] = 1e+4; }
        for (size_t i = nu + _num_latent; i < _num_param; i++) { _lb[i] = 1e-4; _ub[i] = 1e+2; }
        _params = _moihgp->getParams();
        _LBFGSB_param.max_iterations = 1000;                                                        // :100-105
        _LBFGSB_param.m = 10;
        _LBFGSB_param.max_linesearch = 20;
        _LBFGSB_param.ftol = 1e-8;
        _LBFGSB_param.epsilon = 1e-8;
        _LBFGSB_param.epsilon_rel = 1e-8;
        _solver = new opt::LBFGSBSolver(_LBFGSB_param);
        _obj = new RegressionObjective<StateSpace>(_num_data, _moihgp);
    }
    ~MOIHGPRegression() { delete _obj; delete _solver; delete _moihgp; }
    MOIHGPRegression(const MOIHGPRegression&) = delete;
    MOIHGPRegression& operator=(const MOIHGPRegression&) = delete;

    int fit(const std::vector<Vector>& Y) {                                                         // :118-124
        _obj->set_data(Y);
        double fx;
        int num_iter = _solver->minimize(*_obj, _params, fx, _lb, _ub);
        _params = _moihgp->getParams();
        return num_iter;
    }
    std::vector<Vector> predict(const std::vector<Vector>& Y) {                                     // :127-139
        std::vector<Vector> Yhat;
        Yhat.reserve(Y.size());
        Vector x(_num_latent * _dim, 0.0), xnew(x.size()), yhat(_num_output), y(_num_output);
        for (size_t t = 0; t < Y.size(); t++) {
            y = Y[t];
            gp32_step3(_moihgp->handle(), x.data(), y.data(), xnew.data(), yhat.data());
            Yhat.push_back(yhat);
            x = xnew;
        }
        return Yhat;
    }
    Vector getParams() { return _moihgp->getParams(); }
    size_t getNumParam() { return _num_param; }
    size_t getNumOutput() { return _num_output; }
    size_t getNumLatent() { return _num_latent; }
    size_t getNumIGPParam() { return _igp_num_param; }
    size_t getIGPDim() { return _dim; }
    size_t getNumData() { return _num_data; }
    RegressionObjective<StateSpace>& objective() { return *_obj; }   // (not in the reference)

private:
    MOIHGP<StateSpace>* _moihgp;
    bool _threading;
    double _dt;
    size_t _num_output, _num_latent, _num_data, _num_param, _igp_num_param, _dim;
    Vector _params, _lb, _ub;
    opt::LBFGSBParam _LBFGSB_param;
    opt::LBFGSBSolver* _solver;
    RegressionObjective<StateSpace>* _obj;
};

}  // namespace moihgp

#endif
