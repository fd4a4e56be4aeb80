// MistralWaterNative.cs -- the P/Invoke layer over libmistral_water.so (include/mistral_water.h, MW_ABI_VERSION 4).
//
// Drop into Assets/Mistral Water/Scripts/ next to the two MonoBehaviours of this folder; the shared object goes to
// Assets/Plugins/x86_64/libmistral_water.so.  Every extern below mirrors one prototype of the header, argument for
// argument; tests/test_csharp_binding.py parses this file and checks names, argument counts and kinds, struct field
// order and sizes against the header, because no C# toolchain exists in the build image to compile it.
//
// Marshalling: Vector2 / Vector3 / Vector4 / Color are blittable sequential float structs, so a managed array of them is
// pinned by the marshaller and arrives as float*; IntPtr carries device pointers, streams and opaque handles.
using System;
using System.Runtime.InteropServices;
using UnityEngine;

public static class MistralWaterNative
{
    const string Lib = "mistral_water";

    public const int AbiVersion = 4;
    public const int CommIdBytes = 128;

    public enum Status { OK = 0, EINVAL = 1, ENOTPOW2 = 2, ENOTCOMMENSURATE = 3, ENOMEM = 4, EDEVICE = 5, ESTATE = 6 }
    public enum Semantics { FFTMesh = 0, OceanRenderer = 1 }
    public enum PondMode { Wave = 0, Gerstner = 1, GerstnerLevelOne = 2 }
    public const uint OutWhiteScalar = 0u, OutColorRgba = 1u;

    [StructLayout(LayoutKind.Sequential)]   // mw_params: 56 bytes
    public struct Params
    {
        public int resolution;
        public float unit_width;
        public float length;
        public float wind_x;
        public float wind_y;
        public float amplitude;
        public float choppiness;
        public float gravity;
        public float t_division;
        public float mult;
        public ulong seed;
        public int semantics;
        public int device;
    }

    [StructLayout(LayoutKind.Sequential)]   // mw_pond_params: 72 bytes
    public struct PondParams
    {
        public int mode;
        public float amplitude;
        public float frequency;
        public float speed;
        public float steepness;
        public float smoothing;
        public Vector4 wspeed;
        public Vector4 dir_ab;
        public Vector4 dir_cd;
    }

    // ---- library ------------------------------------------------------------------------------------------------
    [DllImport(Lib)] public static extern int mw_abi_version();
    [DllImport(Lib)] public static extern IntPtr mw_build_id();
    [DllImport(Lib)] public static extern IntPtr mw_last_error();
    [DllImport(Lib)] public static extern int mw_device_count();
    [DllImport(Lib)] public static extern void mw_params_default(ref Params p, int semantics);

    // ---- lifecycle ----------------------------------------------------------------------------------------------
    [DllImport(Lib)] public static extern Status mw_ocean_create(ref Params p, out IntPtr ocean);
    [DllImport(Lib)] public static extern void mw_ocean_destroy(IntPtr ocean);
    [DllImport(Lib)] public static extern Status mw_ocean_create_batch(ref Params p, int ntiles, out IntPtr ocean);
    [DllImport(Lib)] public static extern int mw_ocean_batch_size(IntPtr ocean);
    [DllImport(Lib)] public static extern Status mw_ocean_set_stream(IntPtr ocean, IntPtr hipStream);
    [DllImport(Lib)] public static extern Status mw_ocean_use_own_stream(IntPtr ocean);
    [DllImport(Lib)] public static extern IntPtr mw_ocean_get_stream(IntPtr ocean);
    [DllImport(Lib)] public static extern Status mw_ocean_synchronize(IntPtr ocean);
    [DllImport(Lib)] public static extern Status mw_ocean_set_choppiness(IntPtr ocean, float choppiness);

    // ---- spectrum and state -------------------------------------------------------------------------------------
    [DllImport(Lib)] public static extern Status mw_ocean_set_spectrum(IntPtr ocean, Vector2[] h0, Vector2[] h0conj);
    [DllImport(Lib)] public static extern Status mw_ocean_get_spectrum(IntPtr ocean, [Out] Vector2[] h0, [Out] Vector2[] h0conj);
    [DllImport(Lib)] public static extern Status mw_ocean_reinit_spectrum(IntPtr ocean, float length, float windX, float windY, float amplitude, ulong seed);
    [DllImport(Lib)] public static extern Status mw_ocean_get_phase(IntPtr ocean, [Out] float[] phase);
    [DllImport(Lib)] public static extern Status mw_ocean_set_phase(IntPtr ocean, float[] phase);
    [DllImport(Lib)] public static extern Status mw_ocean_set_timer(IntPtr ocean, float timer);
    [DllImport(Lib)] public static extern float mw_ocean_normal_length(IntPtr ocean);
    [DllImport(Lib)] public static extern Status mw_ocean_set_normal_length(IntPtr ocean, float normalLength);
    [DllImport(Lib)] public static extern float mw_ocean_timer(IntPtr ocean);
    [DllImport(Lib)] public static extern Status mw_ocean_reset_timer(IntPtr ocean);

    // ---- mesh ---------------------------------------------------------------------------------------------------
    [DllImport(Lib)] public static extern Status mw_ocean_rest_mesh(IntPtr ocean, [Out] Vector3[] vertices, [Out] Vector3[] normals, [Out] Vector2[] uvs, [Out] int[] indices);
    [DllImport(Lib)] public static extern long mw_ocean_index_count(IntPtr ocean);
    [DllImport(Lib)] public static extern int mw_ocean_grid_size(IntPtr ocean);

    // ---- FFTMesh.EvaluateWaves ----------------------------------------------------------------------------------
    [DllImport(Lib)] public static extern Status mw_ocean_evaluate(IntPtr ocean, float t, [Out] Vector3[] vertices, [Out] Vector3[] normals, [Out] Color[] colors);
    [DllImport(Lib)] public static extern Status mw_ocean_update(IntPtr ocean, float deltaTime, [Out] Vector3[] vertices, [Out] Vector3[] normals, [Out] Color[] colors);
    [DllImport(Lib)] public static extern Status mw_ocean_evaluate_device(IntPtr ocean, float[] t, int nsteps, IntPtr dVertices, IntPtr dNormals, IntPtr dWhite, uint flags);
    [DllImport(Lib)] public static extern int mw_ocean_max_batch(IntPtr ocean);

    // ---- OceanRenderer.GenerateTexture --------------------------------------------------------------------------
    [DllImport(Lib)] public static extern Status mw_ocean_generate_texture(IntPtr ocean, float deltaTime, [Out] float[] height, [Out] Vector2[] dispXZ, [Out] Vector3[] normal, [Out] float[] white);
    [DllImport(Lib)] public static extern Status mw_ocean_generate_texture_device(IntPtr ocean, float deltaTime, IntPtr dHeight, IntPtr dDispXZ, IntPtr dNormal, IntPtr dWhite);
    [DllImport(Lib)] public static extern Status mw_ocean_generate_texture_rgba(IntPtr ocean, float deltaTime, [Out] Color[] height, [Out] Color[] displacement, [Out] Color[] normal, [Out] Color[] white);
    [DllImport(Lib)] public static extern Status mw_ocean_generate_texture_rgba_device(IntPtr ocean, float deltaTime, IntPtr dHeight, IntPtr dDisplacement, IntPtr dNormal, IntPtr dWhite);
    // nframes consecutive GenerateTexture() calls in one enqueue (bit-identical to nframes single calls, the phase included); device destinations [nframes][...]
    [DllImport(Lib)] public static extern Status mw_ocean_generate_texture_steps_device(IntPtr ocean, float[] deltaTime, int nframes, IntPtr dHeight, IntPtr dDispXZ, IntPtr dNormal, IntPtr dWhite);
    [DllImport(Lib)] public static extern Status mw_ocean_generate_texture_steps_rgba_device(IntPtr ocean, float[] deltaTime, int nframes, IntPtr dHeight, IntPtr dDisplacement, IntPtr dNormal, IntPtr dWhite);
    [DllImport(Lib)] public static extern Status mw_ocean_generate_texture_steps(IntPtr ocean, float[] deltaTime, int nframes, [Out] float[] height, [Out] Vector2[] dispXZ, [Out] Vector3[] normal, [Out] float[] white);
    [DllImport(Lib)] public static extern Status mw_ocean_generate_texture_steps_rgba(IntPtr ocean, float[] deltaTime, int nframes, [Out] Color[] height, [Out] Color[] displacement, [Out] Color[] normal, [Out] Color[] white);
    [DllImport(Lib)] public static extern int mw_ocean_max_frames(IntPtr ocean);
    [DllImport(Lib)] public static extern Status mw_ocean_advance_phase(IntPtr ocean, float[] deltaTime, int nframes);
    [DllImport(Lib)] public static extern Status mw_ocean_frame_textures(IntPtr ocean, int frame, out IntPtr dHeight, out IntPtr dDispXZ, out IntPtr dNormal, out IntPtr dWhite);
    [DllImport(Lib)] public static extern Status mw_ocean_displace_mesh(IntPtr ocean, [Out] Vector3[] vertices, [Out] Vector3[] normals, [Out] float[] colors);
    [DllImport(Lib)] public static extern Status mw_ocean_displace_mesh_device(IntPtr ocean, IntPtr dVertices, IntPtr dNormals, IntPtr dColors);

    // ---- page-locked output arrays ------------------------------------------------------------------------------
    [DllImport(Lib)] public static extern Status mw_host_register(IntPtr ptr, UIntPtr bytes);
    [DllImport(Lib)] public static extern Status mw_host_unregister(IntPtr ptr);

    // ---- independent tiles on several devices, RCCL gather ------------------------------------------------------
    [DllImport(Lib)] public static extern Status mw_comm_unique_id([Out] byte[] id);
    [DllImport(Lib)] public static extern Status mw_tiles_create(ref Params p, int ntiles, int[] devices, int maxSteps, out IntPtr tiles);
    [DllImport(Lib)] public static extern Status mw_tiles_create_rank(ref Params p, int device, int maxSteps, byte[] commId, int rank, int nranks, out IntPtr tiles);
    [DllImport(Lib)] public static extern void mw_tiles_destroy(IntPtr tiles);
    [DllImport(Lib)] public static extern int mw_tiles_count(IntPtr tiles);
    [DllImport(Lib)] public static extern int mw_tiles_local_count(IntPtr tiles);
    [DllImport(Lib)] public static extern IntPtr mw_tiles_ocean(IntPtr tiles, int localK);
    [DllImport(Lib)] public static extern Status mw_tiles_evaluate(IntPtr tiles, float[] times, int nsteps, uint flags);
    // the outputs of the LATEST mw_tiles_evaluate: a tile that is gathered owns two output sets used alternately -- ask again after every evaluate once mw_tiles_gather is in use
    [DllImport(Lib)] public static extern Status mw_tiles_outputs(IntPtr tiles, int localK, out IntPtr dVertices, out IntPtr dNormals, out IntPtr dWhite);
    [DllImport(Lib)] public static extern Status mw_tiles_generate_texture(IntPtr tiles, float deltaTime);
    [DllImport(Lib)] public static extern Status mw_tiles_textures(IntPtr tiles, int localK, out IntPtr dHeight, out IntPtr dDispXZ, out IntPtr dNormal, out IntPtr dWhite);
    [DllImport(Lib)] public static extern Status mw_tiles_generate_texture_steps(IntPtr tiles, float[] deltaTime, int nframes);
    [DllImport(Lib)] public static extern Status mw_tiles_frames(IntPtr tiles, int localK, out IntPtr dHeight, out IntPtr dDispXZ, out IntPtr dNormal, out IntPtr dWhite);
    [DllImport(Lib)] public static extern Status mw_tiles_gather(IntPtr tiles, int step, int root);
    [DllImport(Lib)] public static extern Status mw_tiles_gathered(IntPtr tiles, out IntPtr dGathered, out long floatsPerTile);
    [DllImport(Lib)] public static extern Status mw_tiles_synchronize(IntPtr tiles);

    // ---- pond ---------------------------------------------------------------------------------------------------
    [DllImport(Lib)] public static extern Status mw_gerstner_displace(Vector3[] pos, long nverts, Vector3[] waves, int nwaves, float amplitude, float frequency, float steepness, float t, [Out] Vector3[] outPos, int device);
    [DllImport(Lib)] public static extern Status mw_gerstner_displace_device(IntPtr dPos, long nverts, Vector3[] waves, int nwaves, float amplitude, float frequency, float steepness, float t, IntPtr dOut, IntPtr hipStream);
    [DllImport(Lib)] public static extern Status mw_gerstner_displace_steps_device(IntPtr dPos, long nverts, Vector3[] waves, int nwaves, float amplitude, float frequency, float steepness, float[] t, int nsteps, IntPtr dOut, IntPtr hipStream);
    [DllImport(Lib)] public static extern int mw_gerstner_max_steps(int nwaves);
    [DllImport(Lib)] public static extern Status mw_pond_displace(ref PondParams p, Vector3[] pos, long nverts, float t, [Out] Vector3[] outPos, [Out] Vector3[] outNormal, int device);
    [DllImport(Lib)] public static extern Status mw_pond_displace_device(ref PondParams p, IntPtr dPos, long nverts, float t, IntPtr dOut, IntPtr dOutNormal, IntPtr hipStream);

    // ---- helpers ------------------------------------------------------------------------------------------------
    public static string LastError() { return Marshal.PtrToStringAnsi(mw_last_error()); }

    public static void Check(Status s)
    {
        if (s != Status.OK) throw new InvalidOperationException("libmistral_water: " + s + ": " + LastError());
    }

    /// Page-locks a managed array for as long as the returned handle lives (mw_host_register): the per-frame copy of the
    /// results into it then runs at PCIe rate instead of the pageable ~9 GB/s.  Release with Unpin.
    public static GCHandle Pin(Array a, int bytes)
    {
        GCHandle h = GCHandle.Alloc(a, GCHandleType.Pinned);
        Check(mw_host_register(h.AddrOfPinnedObject(), (UIntPtr)(ulong)bytes));
        return h;
    }

    public static void Unpin(GCHandle h)
    {
        if (!h.IsAllocated) return;
        mw_host_unregister(h.AddrOfPinnedObject());
        h.Free();
    }
}
