// OceanRenderer.cs -- drop-in for Assets/Mistral Water/Scripts/OceanRenderer.cs: the same component name, Inspector
// fields (reference :10-27, the seven Shader slots kept so that existing scenes deserialize, but unused) and Unity
// messages (Awake / Update, :76-110).  SetParams + RenderInitial + the 45 blits of GenerateTexture (:116-316) are
// libmistral_water.so in OceanRenderer semantics; the four result textures reach the ocean material under the same
// property names (_Height, _Anim, _Bump, _White, :310-313).  Written against include/mistral_water.h; nothing of the
// reference's method bodies is kept.  tests/test_csharp_binding.py checks every native call of this file.
using System;
using UnityEngine;
using Native = MistralWaterNative;

public class OceanRenderer : MonoBehaviour
{
    // ---- Inspector fields: names, types, ranges and defaults of the reference ----------------------------------
    [Range(0f, 3f)]
    public float mult = 2f;
    public float unitWidth = 1f;
    public int resolution = 256;
    public float length = 256f;
    [Range(0f, 2f)]
    public float choppiness = 1.5f;
    [Range(0f, 2f)]
    public float amplitude = 1f;
    public Vector2 wind;

    public Shader initialShader;
    public Shader spectrumShader;
    public Shader spectrumHeightShader;
    public Shader fftShader;
    public Shader dispersionShader;
    public Shader normalShader;
    public Shader whiteShader;

    // ---- additions ----------------------------------------------------------------------------------------------
    public ulong seed = 1;    // the reference draws _RandomSeed1/2 from the unseeded UnityEngine.Random once, in SetParams
    public int device = 0;

    IntPtr ocean = IntPtr.Zero;
    float oldLength, oldAmplitude;
    Vector2 oldWind;
    Mesh mesh;
    MeshFilter filter;
    Material oceanMat;
    Texture2D heightTexture, displacementTexture, normalTexture, whiteTexture;
    Color[] heightPixels, displacementPixels, normalPixels, whitePixels;
    bool bound = false;

    void Awake()
    {
        filter = GetComponent<MeshFilter>();
        if (filter == null) filter = gameObject.AddComponent<MeshFilter>();
        mesh = new Mesh();
        mesh.indexFormat = UnityEngine.Rendering.IndexFormat.UInt32;
        filter.mesh = mesh;
        oceanMat = GetComponent<MeshRenderer>().material;

        // SetParams + RenderInitial
        Native.Params p = new Native.Params();
        Native.mw_params_default(ref p, (int)Native.Semantics.OceanRenderer);
        p.resolution = resolution;
        p.unit_width = unitWidth;
        p.length = length;
        p.wind_x = wind.x;
        p.wind_y = wind.y;
        p.amplitude = amplitude;
        p.choppiness = choppiness;
        p.mult = mult;
        p.seed = seed;
        p.device = device;
        Native.Check(Native.mw_ocean_create(ref p, out ocean));
        oldLength = length;
        oldAmplitude = amplitude;
        oldWind = wind;

        // GenerateMesh: the resolution x resolution grid
        int n = resolution * resolution;
        Vector3[] vertices = new Vector3[n], normals = new Vector3[n];
        Vector2[] uvs = new Vector2[n];
        int[] indices = new int[(int)Native.mw_ocean_index_count(ocean)];
        Native.Check(Native.mw_ocean_rest_mesh(ocean, vertices, normals, uvs, indices));
        mesh.vertices = vertices;
        mesh.SetIndices(indices, MeshTopology.Triangles, 0);
        mesh.normals = normals;
        mesh.uv = uvs;

        int m = Native.mw_ocean_grid_size(ocean);   // 8 * resolution
        heightTexture = NewTarget(m);
        displacementTexture = NewTarget(m);
        normalTexture = NewTarget(m);
        whiteTexture = NewTarget(m);
        heightPixels = new Color[m * m];
        displacementPixels = new Color[m * m];
        normalPixels = new Color[m * m];
        whitePixels = new Color[m * m];
    }

    static Texture2D NewTarget(int m)
    {
        Texture2D t = new Texture2D(m, m, TextureFormat.RGBAFloat, false, true);
        t.wrapMode = TextureWrapMode.Clamp;
        t.filterMode = FilterMode.Bilinear;
        return t;
    }

    void Update()
    {
        GenerateTexture();
        // the values the materials carry into the NEXT frame, in the reference's order
        Native.Check(Native.mw_ocean_set_choppiness(ocean, choppiness));
        if (oldLength != length || oldWind != wind || oldAmplitude != amplitude)
        {
            // RenderInitial() again with the same seeds: the phase textures keep running
            Native.Check(Native.mw_ocean_reinit_spectrum(ocean, length, wind.x, wind.y, amplitude, seed));
            oldLength = length;
            oldAmplitude = amplitude;
            oldWind = wind;
        }
    }

    // Not in the reference: deltaTimes.Length consecutive GenerateTexture() calls in ONE enqueue (a recorder, an offline bake, a server that
    // produces frames ahead of display): frame k advances the phase by deltaTimes[k] * mult on top of frame k - 1, exactly as the per-frame
    // calls would -- the results are bit-identical to them -- at about half the time per frame.  The ARGBFloat frames land in the caller's
    // DEVICE buffers ([n][M*M*4] floats each; IntPtr.Zero skips a target); the component's own textures are not touched.
    public void GenerateTexturesToDevice(float[] deltaTimes, IntPtr dHeight, IntPtr dDisplacement, IntPtr dNormal, IntPtr dWhite)
    {
        if (deltaTimes.Length > Native.mw_ocean_max_frames(ocean)) throw new ArgumentException("more frames than mw_ocean_max_frames");
        Native.Check(Native.mw_ocean_generate_texture_steps_rgba_device(ocean, deltaTimes, deltaTimes.Length, dHeight, dDisplacement, dNormal, dWhite));
    }

    // the same into managed arrays ([n * M * M] Color each; null skips a target): n frames over PCIe in one synchronous call
    public void GenerateTextures(float[] deltaTimes, Color[] height, Color[] displacement, Color[] normal, Color[] white)
    {
        if (deltaTimes.Length > Native.mw_ocean_max_frames(ocean)) throw new ArgumentException("more frames than mw_ocean_max_frames");
        Native.Check(Native.mw_ocean_generate_texture_steps_rgba(ocean, deltaTimes, deltaTimes.Length, height, displacement, normal, white));
    }

    void GenerateTexture()
    {
        Native.Check(Native.mw_ocean_generate_texture_rgba(ocean, Time.deltaTime, heightPixels, displacementPixels, normalPixels, whitePixels));
        Upload(heightTexture, heightPixels);
        Upload(displacementTexture, displacementPixels);
        Upload(normalTexture, normalPixels);
        Upload(whiteTexture, whitePixels);
        if (!bound)
        {
            oceanMat.SetTexture("_Anim", displacementTexture);
            oceanMat.SetTexture("_Bump", normalTexture);
            oceanMat.SetTexture("_White", whiteTexture);
            oceanMat.SetTexture("_Height", heightTexture);
            bound = true;
        }
    }

    static void Upload(Texture2D t, Color[] pixels)
    {
        t.SetPixelData(pixels, 0);
        t.Apply(false, false);
    }

    /// Checkpoint of the animation: initialTexture, the phase texture and the length the normal pass uses.  The last one differs
    /// from `length` after a length change: the reference sets normalMat._Length once in SetParams (S/OceanRenderer.cs:163).
    public void SaveState(Vector2[] h0, Vector2[] h0conj, float[] phase, out float normalLength)
    {
        Native.Check(Native.mw_ocean_get_spectrum(ocean, h0, h0conj));
        Native.Check(Native.mw_ocean_get_phase(ocean, phase));
        normalLength = Native.mw_ocean_normal_length(ocean);
    }

    public void RestoreState(Vector2[] h0, Vector2[] h0conj, float[] phase, float normalLength)
    {
        Native.Check(Native.mw_ocean_set_spectrum(ocean, h0, h0conj));
        Native.Check(Native.mw_ocean_set_phase(ocean, phase));
        Native.Check(Native.mw_ocean_set_normal_length(ocean, normalLength));
    }

    void OnDestroy()
    {
        if (ocean != IntPtr.Zero)
        {
            Native.mw_ocean_destroy(ocean);
            ocean = IntPtr.Zero;
        }
    }
}
