#pragma once
#include <cstdint>
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

    void prepare(Image& bmp, double blur);   // Heightfield::FillProperties body (src/heightfield.cpp:102-125)
    void build_highmap();                    // Heightfield::BuildHighMap (src/heightfield.cpp:260-290)
};

}  // namespace qdh
