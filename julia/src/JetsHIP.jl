# Package entry point: `] dev path/to/julia` then `using JetsHIP` (JETSHIP_LIB names libjetship.so if it is not on the loader path).
# The binding itself is ../JetsHIP.jl -- one file, so that the static checks (tests/test_julia_binding_static.py) read one file.
include(joinpath(@__DIR__, "..", "JetsHIP.jl"))
