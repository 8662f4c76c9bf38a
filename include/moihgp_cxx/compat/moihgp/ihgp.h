// Drop-in for `#include <moihgp/ihgp.h>`: the per-latent IHGP objects live on the device; nothing to declare on the host.
#include "moihgp.h"
