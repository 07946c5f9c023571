// NativeMethods.cs — P/Invoke declarations for libpisceship.so (include/pisces_hip.h).
// Mirrors the convention of the only native binding in Pisces, src/lib/Common.IO/FileCompression.cs:10-35
// (cdecl, int return code, pinned blittable arrays).  Source only: this image has no dotnet/mono to compile it.
using System;
using System.Runtime.InteropServices;

namespace Pisces.Hip
{
    [StructLayout(LayoutKind.Sequential)]
    public struct PiscesHipConfig
    {
        public int AbiVersion, MinBaseCallQuality, NoiseLevel, MaxVariantQscore, MinVariantQscore, VariantQscoreFilter,
            MinCoverage, LowDepthFilter, MinGenotypeQscore, MaxGenotypeQscore, LowGqFilter, StrandBiasModel,
            FilterSingleStrand, IncludeReferenceCalls, EmitZeroCoverageRefs, ExpectStitchedReads, TileLoci, BlockSize;
        public float MinFrequency, VariantFreqFilter, GenotypeMinFreqFilter, TargetLodFrequency, StrandBiasThreshold,
            NoCallFilterThreshold;
        public int RmxnMaxRepeatLength, RmxnMinRepetitions;
        public float RmxnFrequencyLimit;
        public int Reserved0, Reserved1, Reserved2;
    }

    [StructLayout(LayoutKind.Sequential, Pack = 8, Size = 64)]
    public struct PiscesCalledAllele
    {
        public int Position, TotalCoverage, AlleleSupport, ReferenceSupport, NumNoCalls;
        public int CovF, CovR, CovS, SupF, SupR, SupS;
        public int VariantQscore;
        public double StrandBiasScore;
        public int GenotypeQscore;
        public ushort FilterBits, Info;
    }

    [StructLayout(LayoutKind.Sequential)]
    public unsafe struct PiscesReadBatch
    {
        public int NReads;
        public int* Position; public byte* Flags; public int* CigarOffset; public byte* CigarOp; public uint* CigarLen;
        public int* SeqOffset; public byte* Bases; public byte* Quals; public byte* Directions;
    }

    internal static unsafe class NativeMethods
    {
        private const string Lib = "pisceship";   // libpisceship.so next to Pisces.dll (like libFileCompression.so)

        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_abi_version();
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_default_config(out PiscesHipConfig cfg);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_create(ref PiscesHipConfig cfg, int device, out IntPtr handle);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_destroy(IntPtr handle);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern IntPtr pisces_hip_last_error(IntPtr handle);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_set_reference(IntPtr handle, byte[] upperBases, long length);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_set_intervals(IntPtr handle, int[] starts, int[] ends, int n);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_add_reads(IntPtr handle, ref PiscesReadBatch batch);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_flush(IntPtr handle, int upToPosition, [Out] PiscesCalledAllele[] output, long capacity, out long nOut);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_get_counts(IntPtr handle, int startPosition, int n, [Out] int[] counts);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_add_gapped_mnv_ref(IntPtr handle, int[] positions, int[] counts, int n);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_stats(IntPtr handle, [Out] long[] stats4);

        public static void Check(IntPtr handle, int rc)
        {
            if (rc == 0) return;
            var msg = Marshal.PtrToStringAnsi(pisces_hip_last_error(handle));
            // surfaces through BaseGenomeProcessor's per-job catch (Pisces.Processing/Logic/BaseGenomeProcessor.cs:121-128)
            if (rc == -1 || rc == -4) throw new ArgumentException(msg);
            throw new Exception("libpisceship error " + rc + ": " + msg);
        }
    }
}
