// Renderable.swift:72-87, Primitives.swift, CSG.swift and ConstantDensityVolume.swift of the reference: every
// renderable describes itself to the library as a shape (children first, then the operation) and gets a shape id
// back.  The reference needs one hand-written intersection function per combination of generics
// (Shaders.metal:254-341, names derived at Renderable.swift:157-165); here any nesting of up to 64 leaf primitives
// under one object goes through the same evaluator, so `Subtract<Intersection<Cuboid, Sphere>, Union<...>>` needs
// nothing registered anywhere.
import CRayTracerAMD

public protocol Renderable {
    /// asImpl(&encoder) of the reference: writes the materials, then the shape; returns the shape's id.
    func addShape(_ encoder: inout MaterialEncoder, _ scene: OpaquePointer) throws -> Int32
}

public struct Sphere: Renderable {
    public var transform: Transform
    public var radius: Float
    public var material: any Material

    public init(transform: Transform, radius: Float, material: any Material) {
        self.transform = transform; self.radius = radius; self.material = material
    }
    public init(center: Vec3, radius: Float, material: any Material) {
        self.init(transform: Transform(translation: center), radius: radius, material: material)
    }

    public func addShape(_ e: inout MaterialEncoder, _ sc: OpaquePointer) throws -> Int32 {
        var rec = RtSphere(transform: transform.record, radius: radius, material_offset: try material.encode(into: &e))
        return try check(rt_shape_sphere(sc, &rec))
    }
}

public struct Cylinder: Renderable {
    public var transform: Transform
    public var radius: Float, height: Float
    public var bottom: any Material, top: any Material, side: any Material

    public init(transform: Transform = .init(), radius: Float, height: Float, bottom: any Material, top: any Material, side: any Material) {
        self.transform = transform; self.radius = radius; self.height = height
        self.bottom = bottom; self.top = top; self.side = side
    }
    public init(transform: Transform = .init(), radius: Float, height: Float, material: any Material) {
        self.init(transform: transform, radius: radius, height: height, bottom: material, top: material, side: material)
    }

    public func addShape(_ e: inout MaterialEncoder, _ sc: OpaquePointer) throws -> Int32 {
        // three records, in the reference's order (Primitives.swift:133-145)
        var rec = RtCylinder(transform: transform.record, radius: radius, height: height,
                             bottom_material_offset: try bottom.encode(into: &e), top_material_offset: try top.encode(into: &e),
                             side_material_offset: try side.encode(into: &e))
        return try check(rt_shape_cylinder(sc, &rec))
    }
}

public struct Cuboid: Renderable {
    public var transform: Transform
    public var size: Vec3
    public var material: any Material

    public init(transform: Transform = .init(), size: Vec3, material: any Material) {
        self.transform = transform; self.size = size; self.material = material
    }

    public func addShape(_ e: inout MaterialEncoder, _ sc: OpaquePointer) throws -> Int32 {
        let m = try material.encode(into: &e)       // one record, its handle on all six faces (Primitives.swift:186-193)
        var rec = RtCuboid(transform: transform.record, size: RtFloat3(size), material_offset: (m, m, m, m, m, m))
        return try check(rt_shape_cuboid(sc, &rec))
    }
}

public struct Quad: Renderable {
    public var origin: Vec3, u: Vec3, v: Vec3
    public var material: any Material

    public init(origin: Vec3, u: Vec3, v: Vec3, material: any Material) {
        self.origin = origin; self.u = u; self.v = v; self.material = material
    }

    public func addShape(_ e: inout MaterialEncoder, _ sc: OpaquePointer) throws -> Int32 {
        // Primitives.swift:266-274: n = u x v, w = n / |n|^2, normal = n / |n|, d = origin . normal
        let n = Vec3(u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x)
        let len2 = (n * n).sum()
        let normal = n / len2.squareRoot()
        var rec = RtQuad(origin: RtFloat3(origin), u: RtFloat3(u), v: RtFloat3(v), w: RtFloat3(n / len2), normal: RtFloat3(normal),
                         d: (origin * normal).sum(), material_offset: try material.encode(into: &e))
        return try check(rt_shape_quad(sc, &rec))
    }
}

private func csg(_ op: RtCsgOp, _ children: [any Renderable], _ e: inout MaterialEncoder, _ sc: OpaquePointer) throws -> Int32 {
    var ids: [Int32] = []
    for c in children { ids.append(try c.addShape(&e, sc)) }
    return try check(rt_shape_csg(sc, Int32(op.rawValue), Int32(ids.count), &ids))
}

/// CSG.swift:17-35.
public struct Subtract<LHS: Renderable, RHS: Renderable>: Renderable {
    public var lhs: LHS, rhs: RHS
    public init(lhs: LHS, rhs: RHS) { self.lhs = lhs; self.rhs = rhs }
    public func addShape(_ e: inout MaterialEncoder, _ sc: OpaquePointer) throws -> Int32 { try csg(RT_CSG_SUBTRACT, [lhs, rhs], &e, sc) }
}

/// CSG.swift:53-77.
public struct Union<each T: Renderable>: Renderable {
    public var items: (repeat each T)
    public init(_ item: repeat each T) { items = (repeat each item) }
    public func addShape(_ e: inout MaterialEncoder, _ sc: OpaquePointer) throws -> Int32 {
        var all: [any Renderable] = []
        repeat all.append(each items)
        return try csg(RT_CSG_UNION, all, &e, sc)
    }
}

/// CSG.swift:95-119.
public struct Intersection<each T: Renderable>: Renderable {
    public var items: (repeat each T)
    public init(_ item: repeat each T) { items = (repeat each item) }
    public func addShape(_ e: inout MaterialEncoder, _ sc: OpaquePointer) throws -> Int32 {
        var all: [any Renderable] = []
        repeat all.append(each items)
        return try csg(RT_CSG_INTERSECTION, all, &e, sc)
    }
}

/// ConstantDensityVolume.swift:17-33: the boundary's material is the medium's phase function (ColoredIsotropic).
public struct ConstantDensityVolume<Base: Renderable>: Renderable {
    public var base: Base
    public var density: Float
    public init(base: Base, density: Float) { self.base = base; self.density = density }
    public func addShape(_ e: inout MaterialEncoder, _ sc: OpaquePointer) throws -> Int32 {
        try check(rt_shape_cdv(sc, try base.addShape(&e, sc), density))
    }
}

/// Scene.swift:9-16.
public struct Scene {
    public var camera: CameraConfig
    public var objects: [any Renderable]
    public init(camera: CameraConfig, objects: [any Renderable]) { self.camera = camera; self.objects = objects }
}
