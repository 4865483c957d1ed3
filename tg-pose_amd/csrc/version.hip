#include "tgp_common.h"
extern "C" int tgp_version(void) { return TGP_ABI_VERSION; }
