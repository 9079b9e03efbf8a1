from conformer_amd.model.utils.convolution import (ConvolutionModule, ConvolutionSubsampling,  # noqa: F401
                                                    DepthWiseSeperableConvolution, DownsamplingConvolution)
