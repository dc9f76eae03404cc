// SceneBuffers.swift:9-72 and the dispatch half of Renderer (ContentView.swift:122-224) of the reference over the
// C ABI.  Device memory comes from the caller (hipMalloc through any HIP binding, or a buffer shared with a display
// API); this file allocates nothing on the device.
import CRayTracerAMD

/// SceneBuffers: the flattened scene.  The reference sizes a material buffer, encodes every material, groups the
/// renderables by concrete type and builds the acceleration structure; the library does all of it in rt_scene_commit.
public final class SceneBuffers {
    let context: OpaquePointer
    let scene: OpaquePointer

    public init(scene description: Scene, device: Int32 = 0, kitConventions: Bool = false) throws {
        var ctx: OpaquePointer?
        try check(rt_context_create(device, &ctx))
        context = ctx!
        var sc: OpaquePointer?
        try check(rt_scene_create(context, &sc))
        scene = sc!
        if kitConventions {      // scenes of the RayTracingKit command-line renderer (its texture coordinates, its metal rule)
            try check(rt_scene_set_option(scene, Int32(RT_OPTION_CONVENTION.rawValue), Int32(RT_CONVENTION_KIT.rawValue)))
        }
        var encoder = MaterialEncoder(scene: scene)
        for object in description.objects {              // RenderableGrouper.add (SceneBuffers.swift:116-119)
            try check(rt_add_object(scene, try object.addShape(&encoder, scene)))
        }
        try check(rt_scene_commit(scene))                // tree build + upload (SceneBuffers.swift:42-71)
    }

    deinit {
        rt_scene_destroy(scene)
        rt_context_destroy(context)
    }

    public var stats: RtSceneStats {
        var s = RtSceneStats()
        rt_scene_stats(scene, &s)
        return s
    }
}

/// Renderer.draw(in:) without the view: one call adds `samplesPerPixel` samples of every pixel to `accumulator`
/// (width x height x 3 floats in device memory).  `workspace` is device memory of at least
/// `Renderer.workspaceSize(width:height:)` bytes; `stream` is a hipStream_t or nil.
public final class Renderer {
    public let buffers: SceneBuffers
    public var camera: CameraConfig
    public private(set) var passCounter: UInt32 = 0

    public init(scene: Scene, device: Int32 = 0) throws {
        buffers = try SceneBuffers(scene: scene, device: device)
        camera = scene.camera
    }

    public static func workspaceSize(width: Int, height: Int) -> Int {
        rt_render_workspace_size(Int32(RT_PIPELINE_FUSED.rawValue), Int32(width), Int32(height))
    }

    /// The interactive loop of ContentView.swift:182-224: a fresh seed per pass, 1 sample per pixel, depth 10, then the
    /// running mean in the 10-bit accumulator and the bgra8 drawable (Shaders.metal:151-158).
    public func drawInteractive(width: Int, height: Int, passSum: UnsafeMutablePointer<Float>, accumulator10: UnsafeMutablePointer<UInt32>,
                                drawable: UnsafeMutablePointer<UInt8>, workspace: UnsafeMutableRawPointer, workspaceBytes: Int,
                                stream: UnsafeMutableRawPointer? = nil) throws {
        guard passCounter < 200 else { return }          // ContentView.swift:188: pause at 200 passes
        passCounter += 1
        var rng = SystemRandomNumberGenerator()
        var rc = RenderConfig()
        rc.passCounter = passCounter
        rc.rngSeed = rng.next()
        var r = rc.record, c = camera.record
        // passSum must be zero on entry (hipMemsetAsync on the same stream): the present step reads one pass's sums
        try check(rt_render(buffers.scene, &c, &r, 0, Int32(RT_PIPELINE_FUSED.rawValue), Int32(width), Int32(height), passSum, workspace,
                            workspaceBytes, stream))
        try check(rt_progressive_present(buffers.context, passSum, 1, passCounter, accumulator10, drawable, Int32(width), Int32(height), stream))
    }

    /// The command-line renderer's single render (RayTracing/main.swift:131-139): all samples in one call, then the
    /// Kit's gamma and 8-bit quantisation into an RGBX8 image.
    public func render(width: Int, height: Int, samplesPerPixel: UInt32, maxDepth: UInt32, seed: UInt64, accumulator: UnsafeMutablePointer<Float>,
                       rgbx: UnsafeMutablePointer<UInt8>, workspace: UnsafeMutableRawPointer, workspaceBytes: Int,
                       stream: UnsafeMutableRawPointer? = nil) throws {
        var rc = RenderConfig()
        rc.samplesPerPixel = samplesPerPixel
        rc.maxDepth = maxDepth
        rc.rngSeed = seed
        var r = rc.record, c = camera.record
        try check(rt_render(buffers.scene, &c, &r, 0, Int32(RT_PIPELINE_FUSED.rawValue), Int32(width), Int32(height), accumulator, workspace,
                            workspaceBytes, stream))
        try check(rt_resolve_rgb8(buffers.context, accumulator, samplesPerPixel, Int32(width), Int32(height), rgbx, stream))
    }

    /// Camera change or resize: the running mean starts over (ContentView.swift:107-113,172-180).
    public func setNeedsRedraw() { passCounter = 0 }
}
