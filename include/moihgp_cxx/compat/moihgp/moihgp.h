// Drop-in for `#include <moihgp/moihgp.h>` of the reference (moihgp/include/moihgp/moihgp.h): put this directory AHEAD of the
// reference's include directory (-I <repo>/include/moihgp_cxx/compat -I <reference>/moihgp/include) and the reference's own
// C++ clients -- moihgp_online.h, moihgp_regression.h with LBFGS++ and Eigen -- compile against libmoihgp.so (HIP) unchanged:
// moihgp::MOIHGP<moihgp::Matern32StateSpace> keeps its constructor, step / negLogLikelihood overloads, update, getParams and
// getters, with Eigen containers passed straight through (the class is generic over the vector type).  INTEGRATION.md section 1.
#ifndef MOIHGP_COMPAT_MOIHGP_H_
#define MOIHGP_COMPAT_MOIHGP_H_
#include "../../moihgp.hpp"
#endif
