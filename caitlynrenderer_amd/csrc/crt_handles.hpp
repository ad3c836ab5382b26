// Opaque handle types behind include/crt.h, shared by the translation units that create them.
#pragma once
#include "host/cwbvh.hpp"
#include "host/obj_loader.hpp"
#include "host/sbvh.hpp"

struct crt_sbvh { crt::SBVH bvh; };
struct crt_cwbvh { crt::CWBVH bvh; };
struct crt_mesh { crt::Mesh mesh; };
