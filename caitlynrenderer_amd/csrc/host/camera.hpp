// Pinhole camera kept on the host; mirrors Caitlyn/Camera.h:4-66 (position/lookAt/fov in
// degrees -> yaw/pitch -> forward/right/up).  Only the seven uniforms of
// Caitlyn/Scene.h:1143-1149 cross the C ABI (crt_camera).
#pragma once
#include <cmath>

#include "../../../include/crt.h"
#include "vecmath.hpp"

namespace crt {

struct Camera {
    float3 position, up, right, forward, worldUp{0.f, 1.f, 0.f};
    float pitch = 0.f, yaw = 0.f, fov = 0.f, focalDist = 0.1f, aperture = 0.f;
    bool isMoving = false;

    Camera() = default;
    Camera(float3 pos, float3 lookAt, float fovDeg) {           // Camera.h:7-19
        position = pos;
        float3 dir = normalize(lookAt - position);
        pitch = degrees(std::asin(dir.y));
        yaw = degrees(std::atan2(dir.z, dir.x));
        fov = radians(fovDeg);
        updateCamera();
    }
    void offsetOrientation(float x, float y) { pitch -= y; yaw += x; updateCamera(); isMoving = true; }   // Camera.h:35-40
    void offsetPosition(float3 d) { position += d; updateCamera(); isMoving = true; }                      // Camera.h:42-46
    void updateCamera() {                                        // Camera.h:48-58
        float3 f;
        f.x = std::cos(radians(yaw)) * std::cos(radians(pitch));
        f.y = std::sin(radians(pitch));
        f.z = std::sin(radians(yaw)) * std::cos(radians(pitch));
        forward = normalize(f);
        right = normalize(cross(forward, worldUp));
        up = normalize(cross(right, forward));
    }
    crt_camera abi() const {
        crt_camera c;
        for (int i = 0; i < 3; ++i) { c.position[i] = position[i]; c.right[i] = right[i]; c.up[i] = up[i]; c.forward[i] = forward[i]; }
        c.fov = fov; c.focal_dist = focalDist; c.aperture = aperture;
        return c;
    }
    // glm::radians / glm::degrees for float: x * (pi/180), x * (180/pi) in fp32
    static float radians(float d) { return d * 0.01745329251994329576923690768489f; }
    static float degrees(float r) { return r * 57.295779513082320876798154814105f; }
};

}  // namespace crt
