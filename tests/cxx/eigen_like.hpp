// TEST-ONLY stand-in with the handful of Eigen::VectorXd behaviours the reference's C++ learners rely on
// (moihgp_online.h, moihgp_regression.h): signed size(), unsized default construction, resize, operator[] / operator(),
// lazy `a - b` expressions passed straight into MOIHGP::step, `+=`, `/=`, setZero(n).  It exists to check that
// include/moihgp_cxx/moihgp.hpp really is generic over such a vector library (Eigen itself is not installed here).
#pragma once
#include <cstddef>
#include <vector>

namespace eigen_like {
struct VectorXd;
struct Diff {                       // lazy a - b
    const VectorXd& a; const VectorXd& b;
    double operator[](long i) const;
    double operator()(long i) const { return (*this)[i]; }
    long size() const;
};
struct VectorXd {
    std::vector<double> v;
    VectorXd() {}
    explicit VectorXd(long n) : v((size_t)n, 0.0) {}
    VectorXd(const Diff& d) { *this = d; }
    VectorXd& operator=(const Diff& d) { v.resize((size_t)d.size()); for (long i = 0; i < d.size(); i++) v[(size_t)i] = d[i]; return *this; }
    long size() const { return (long)v.size(); }
    void resize(long n) { v.resize((size_t)n); }
    VectorXd& setZero(long n) { v.assign((size_t)n, 0.0); return *this; }
    VectorXd& setZero() { v.assign(v.size(), 0.0); return *this; }
    double& operator[](long i) { return v[(size_t)i]; }
    double operator[](long i) const { return v[(size_t)i]; }
    double& operator()(long i) { return v[(size_t)i]; }
    double operator()(long i) const { return v[(size_t)i]; }
    VectorXd& operator+=(const VectorXd& o) { for (size_t i = 0; i < v.size(); i++) v[i] += o.v[i]; return *this; }
    VectorXd& operator/=(double s) { for (auto& e : v) e /= s; return *this; }
};
inline double Diff::operator[](long i) const { return a[i] - b[i]; }
inline long Diff::size() const { return a.size(); }
inline Diff operator-(const VectorXd& a, const VectorXd& b) { return Diff{a, b}; }
}  // namespace eigen_like
