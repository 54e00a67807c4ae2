// Compile-time weight / KV-cache quantization policies -- kept verbatim in meaning from
// /root/reference/Mila/Src/Dnn/Quantization/Weight/Policies.ixx:23-31,46-53,70-79,104-113,125-138
// and Quantization/KvCache/Policy.ixx:51-54,75-78.  They carry no runtime state: every branch on
// them is `if constexpr`.
#pragma once

#include <concepts>

#include "Core.h"

namespace Mila::Dnn::Quant::Weight
{
    /// identity policy: no quantization
    struct NoWeightQuant
    {
        static constexpr bool kIsQuantized = false;
        static constexpr TensorDataType kStorageDtype = TensorDataType::FP32;   // sentinel, never read
        static constexpr TensorDataType kScaleDtype = TensorDataType::FP32;
        static constexpr bool kPerChannel = false;
    };

    /// per-output-channel FP8 (bf16 -> e4m3 at load time; scale[o] = max|W[o,:]| / 448)
    template<TensorDataType TStorage = TensorDataType::FP8_E4M3>
    struct PerChannelFp8
    {
        static constexpr bool kIsQuantized = true;
        static constexpr TensorDataType kStorageDtype = TStorage;
        static constexpr TensorDataType kScaleDtype = TensorDataType::FP32;
        static constexpr bool kPerChannel = true;
    };

    /// per-group INT4 (GPTQ checkpoints; no quantize-on-load path in the reference either,
    /// CudaLinearOp.ixx:385-391).  Declared for surface parity; no CDNA4 op row.
    template<int kGroupSize = 128>
    struct PerGroupInt4
    {
        static constexpr bool kIsQuantized = true;
        static constexpr TensorDataType kStorageDtype = TensorDataType::UINT8;
        static constexpr TensorDataType kScaleDtype = TensorDataType::FP32;
        static constexpr bool kPerChannel = false;
        static constexpr int kQuantizationGroupSize = kGroupSize;
        static constexpr bool kIsFp4E2M1 = false;
    };

    /// per-group FP4 E2M1, packed two nibbles per byte (low nibble = even column),
    /// scale[g] = max|W[g,:]| / 6, values {0,.5,1,1.5,2,3,4,6} with bit 3 = sign
    template<int kGroupSize = 128>
    struct PerGroupFp4
    {
        static constexpr bool kIsQuantized = true;
        static constexpr TensorDataType kStorageDtype = TensorDataType::UINT8;
        static constexpr TensorDataType kScaleDtype = TensorDataType::FP32;
        static constexpr bool kPerChannel = false;
        static constexpr int kQuantizationGroupSize = kGroupSize;
        static constexpr bool kIsFp4E2M1 = true;
    };

    template<typename T>
    concept WeightQuantPolicy = requires
    {
        { T::kIsQuantized } -> std::convertible_to<bool>;
        { T::kStorageDtype } -> std::convertible_to<TensorDataType>;
        { T::kScaleDtype } -> std::convertible_to<TensorDataType>;
        { T::kPerChannel } -> std::convertible_to<bool>;
    };

    static_assert( WeightQuantPolicy<NoWeightQuant> );
    static_assert( WeightQuantPolicy<PerChannelFp8<>> );
    static_assert( WeightQuantPolicy<PerGroupInt4<>> );
    static_assert( WeightQuantPolicy<PerGroupFp4<>> );

    /// storage format id of the C ABI (mila_cdna4.h: 0 bf16, 1 fp8 per-channel, 2 fp4 per-group)
    template<WeightQuantPolicy P>
    constexpr int abiWeightFormat()
    {
        if constexpr ( !P::kIsQuantized ) return 0;
        else if constexpr ( P::kPerChannel ) return 1;
        else return 2;
    }

    template<WeightQuantPolicy P>
    constexpr int groupSizeOf()
    {
        if constexpr ( requires { P::kQuantizationGroupSize; } ) return P::kQuantizationGroupSize;
        else return 0;
    }
}

namespace Mila::Dnn::Quant::KvCache
{
    /// unbounded cache: capacity == context length
    struct NoKvCompression { static constexpr bool kBoundedRing = false; };
    /// bounded ring: capacity = min(T, window + prefill_chunk - 1)  (CudaGqaOp.ixx:552-574)
    struct SlidingWindowKvCache { static constexpr bool kBoundedRing = true; };
}
