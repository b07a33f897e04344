// placeholder until the device-resident forest lands
#include "mht_common.h"
namespace mht {
void forest_destroy(mht_ctx*) {}
}
