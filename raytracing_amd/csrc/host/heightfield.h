#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "image.h"
#include "vecmath.h"

namespace qdh {

struct HeightfieldData {
    uint32_t width = 0, height = 0;
    std::vector<float> heights, maxheights, highmap;   // highmap: 16 floats per cell
    std::vector<V3> normals;
    V3 bmin, bmax;
    bool use_optimization = false;
    int max_k = 0;

    // false (with a message) only when a device hook fails
    bool prepare(Image& bmp, double blur, std::string& err);   // Heightfield::FillProperties body (src/heightfield.cpp:102-125)
    bool build_highmap(std::string& err);                      // Heightfield::BuildHighMap (src/heightfield.cpp:260-290)
};

// Device entry points for the table passes (qdh_set_prep_hooks); null members = the CPU code in heightfield.cpp / image.cpp.
struct PrepHooks {
    int (*heightfield)(uint32_t, uint32_t, const float*, const float*, int, float*, float*, double*) = nullptr;
    int (*highmap)(uint32_t, uint32_t, const float*, int, float*) = nullptr;
    int (*differentiate)(uint32_t, uint32_t, float*) = nullptr;
};
extern PrepHooks g_prep_hooks;
extern int g_prep_hook_calls;          // table passes that ran through a hook so far (tests)

}  // namespace qdh
