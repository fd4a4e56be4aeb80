// FFTMesh.cs -- drop-in for Assets/Mistral Water/Scripts/FFTMesh.cs: the same component name, Inspector fields
// (reference :9-23) and Unity messages (Awake / Update, :60-84); the private numerical methods -- the Phillips /
// htilde0 spectrum fill, htilde, the O(N^4) Displacement loop and the Jacobian of EvaluateWaves (:141-280) -- are
// libmistral_water.so on an MI355X.  Written against include/mistral_water.h; nothing of the reference's method
// bodies is kept.  Not compilable in the build image (no C# toolchain, UnityEngine.dll is proprietary):
// tests/test_csharp_binding.py checks every native call of this file against MistralWaterNative.cs.
using System;
using System.Runtime.InteropServices;
using UnityEngine;
using Native = MistralWaterNative;

public class FFTMesh : MonoBehaviour
{
    // ---- Inspector fields: names, types and defaults of the reference ------------------------------------------
    public float choppiness = 1f;
    public float tDivision = 1f;
    public int resolution = 50;
    public float unitWidth = 1f;
    public bool generate = false;
    public float length = 1f;
    public Vector2 wind = new Vector2(1f, 1f);
    public float amplitude = 1f;

    // ---- additions (the reference never seeds UnityEngine.Random; the library's generator is a documented counter RNG)
    public ulong seed = 1;
    public bool fixedSeed = false;   // true: every regeneration reproduces the same sea; false: a new one, like GenerateMesh
    public int device = 0;

    IntPtr ocean = IntPtr.Zero;
    ulong generation = 0;
    MeshFilter filter;
    Mesh mesh;
    Vector3[] vertices, displaced, normals;
    Vector2[] uvs;
    Color[] colors;
    int[] indices;
    GCHandle pinDisplaced, pinNormals, pinColors;
    float timer = 0f;

    void Awake()
    {
        filter = GetComponent<MeshFilter>();
        if (filter == null) filter = gameObject.AddComponent<MeshFilter>();
        mesh = new Mesh();
        mesh.indexFormat = UnityEngine.Rendering.IndexFormat.UInt32;   // 1024^2 vertices do not fit 16-bit indices
        filter.mesh = mesh;
        Regenerate();
    }

    void Update()
    {
        if (generate)
        {
            timer = 0f;
            Regenerate();
            generate = false;
        }
        timer += Time.deltaTime / tDivision;
        EvaluateWaves(timer);
    }

    void OnDestroy() { Release(); }

    // SetParams + GenerateMesh: handle, rest mesh, spectrum (a fresh draw on every regeneration unless fixedSeed)
    void Regenerate()
    {
        Release();
        Native.Params p = new Native.Params();
        Native.mw_params_default(ref p, (int)Native.Semantics.FFTMesh);
        p.resolution = resolution;
        p.unit_width = unitWidth;
        p.length = length;
        p.wind_x = wind.x;
        p.wind_y = wind.y;
        p.amplitude = amplitude;
        p.choppiness = choppiness;
        p.t_division = tDivision;
        p.seed = fixedSeed ? seed : seed + generation;
        p.device = device;
        generation++;
        Native.Check(Native.mw_ocean_create(ref p, out ocean));

        int n = resolution * resolution;
        vertices = new Vector3[n];
        displaced = new Vector3[n];
        normals = new Vector3[n];
        uvs = new Vector2[n];
        colors = new Color[n];
        indices = new int[(int)Native.mw_ocean_index_count(ocean)];
        Native.Check(Native.mw_ocean_rest_mesh(ocean, vertices, normals, uvs, indices));
        mesh.Clear();
        mesh.vertices = vertices;
        mesh.SetIndices(indices, MeshTopology.Triangles, 0);
        mesh.normals = normals;
        mesh.uv = uvs;

        // the three arrays EvaluateWaves fills every frame stay page-locked for the life of the handle
        pinDisplaced = Native.Pin(displaced, n * 12);
        pinNormals = Native.Pin(normals, n * 12);
        pinColors = Native.Pin(colors, n * 16);
    }

    void EvaluateWaves(float t)
    {
        Native.Check(Native.mw_ocean_set_choppiness(ocean, choppiness));   // the reference reads the live field every frame
        Native.Check(Native.mw_ocean_evaluate(ocean, t, displaced, normals, colors));
        mesh.vertices = displaced;
        mesh.normals = normals;
        mesh.colors = colors;
    }

    /// Evaluate the ocean Unity itself generated: pass the reference's own htilde0 draws (verttilde / vertConj).
    public void SetSpectrum(Vector2[] h0, Vector2[] h0conj)
    {
        Native.Check(Native.mw_ocean_set_spectrum(ocean, h0, h0conj));
    }

    void Release()
    {
        Native.Unpin(pinDisplaced);
        Native.Unpin(pinNormals);
        Native.Unpin(pinColors);
        if (ocean != IntPtr.Zero)
        {
            Native.mw_ocean_destroy(ocean);
            ocean = IntPtr.Zero;
        }
    }
}
