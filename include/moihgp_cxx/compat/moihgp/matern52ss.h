// Drop-in for `#include <moihgp/matern52ss.h>`: moihgp::Matern52StateSpace is a tag here (the model lives on the device).
#include "moihgp.h"
