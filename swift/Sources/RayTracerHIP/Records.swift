// Scene API of the reference's MetalRayTracer target (Renderable.swift:34-91, Primitives.swift, CSG.swift,
// ConstantDensityVolume.swift, Material.swift, Config.swift) over the C ABI of include/raytracer_amd.h.
// Not compiled in this repository (no Swift toolchain in the build image); kept thin on purpose.
//
// On Apple platforms the reference's own records (`__Sphere`, `__Cuboid`, `__CameraConfig`, ... imported from
// Renderable.h / Config.h with swift_private) have the byte layout of RtSphere, RtCuboid, RtCameraConfig, ... and can
// be handed to the rt_shape_* calls unchanged.  On Linux there is no <simd/simd.h>, so this file builds the records
// from Swift's own SIMD types, which share the layout: SIMD3<Float> is 16 bytes, 16-byte aligned, x y z first.
import CRayTracerAMD

public typealias Vec3 = SIMD3<Float>

extension RtFloat3 {
    init(_ v: Vec3) { self.init(x: v.x, y: v.y, z: v.z, unused: 0) }
}

/// Unit quaternion + translation (Renderable.swift:34-70).  `rotation` is (imag.x, imag.y, imag.z, real).
public struct Transform {
    public var rotation: SIMD4<Float>
    public var translation: Vec3

    public init(rotation: SIMD4<Float> = .init(0, 0, 0, 1), translation: Vec3 = .zero) {
        self.rotation = rotation
        self.translation = translation
    }

    /// Columns of the rotation matrix of a unit quaternion: what simd_float3x3(simd_quatf) returns.
    var columns: (Vec3, Vec3, Vec3) {
        let x = rotation.x, y = rotation.y, z = rotation.z, w = rotation.w
        return (Vec3(1 - 2 * (y * y + z * z), 2 * (x * y + w * z), 2 * (x * z - w * y)),
                Vec3(2 * (x * y - w * z), 1 - 2 * (x * x + z * z), 2 * (y * z + w * x)),
                Vec3(2 * (x * z + w * y), 2 * (y * z - w * x), 1 - 2 * (x * x + y * y)))
    }

    var record: RtTransform {
        let c = columns
        return RtTransform(rotation: RtMatrix3x3(columns: (RtFloat3(c.0), RtFloat3(c.1), RtFloat3(c.2))),
                           translation: RtFloat3(translation))
    }
}

/// Config.swift:9-70 without the orbit accessors of the app's sliders.
public struct CameraConfig {
    public var verticalFOV: Float = 20
    public var lookFrom: Vec3 = .init(0, 0, 1)
    public var lookAt: Vec3 = .zero
    public var up: Vec3 = .init(0, 1, 0)
    public var defocusAngle: Float = 0
    public var focusDistance: Float = 1
    public var background: RtBackgroundLighting = rt_background_lighting_sky
    public init() {}

    var record: RtCameraConfig {
        RtCameraConfig(vertical_FOV: verticalFOV, look_from: RtFloat3(lookFrom), look_at: RtFloat3(lookAt), up: RtFloat3(up),
                       defocus_angle: defocusAngle, focus_distance: focusDistance, background: background.rawValue)
    }
}

/// Config.swift:72-88.
public struct RenderConfig {
    public var samplesPerPixel: UInt32 = 1
    public var maxDepth: UInt32 = 10
    public var passCounter: UInt32 = 1
    public var rngSeed: UInt64 = 0
    public init() {}

    var record: RtRenderConfig {
        RtRenderConfig(samples_per_pixel: samplesPerPixel, max_depth: maxDepth, pass_counter: passCounter, rng_seed: rngSeed)
    }
}

public struct RayTracerError: Error, CustomStringConvertible {
    public let code: Int32
    public let description: String
}

/// Turns a negative return code of the C ABI into a thrown error carrying rt_last_error().
@discardableResult
func check(_ code: Int32) throws -> Int32 {
    if code < 0 { throw RayTracerError(code: code, description: String(cString: rt_last_error())) }
    return code
}
