// Drop-in for `#include <moihgp/matern32ss.h>`: moihgp::Matern32StateSpace is a tag here (the model lives on the device).
#include "moihgp.h"
