// .3ds reader of the host layer (load_3ds.cc): what the reference's loader takes out of lib3ds (Loader.cc:276-353).
#pragma once
#include <cstdint>
#include <vector>

namespace mi355 {

struct Corner3ds { float pos[3]; float normal[3]; };                 // one per triangle corner, in face order
struct Face3ds { uint32_t r, g, b, two_sided; };                     // diffuse colour bytes of the face's material

// throws std::string on malformed or unsupported input
void load3ds(const std::vector<unsigned char> &file_bytes, std::vector<Corner3ds> &corners, std::vector<Face3ds> &faces);

} // namespace mi355
