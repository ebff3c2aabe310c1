# VecScatter.cmake -- the reference's VSTARGET(name input output) hook (cmake/VecScatter.cmake:10-31 of bricklib) for the
# MI355X engine: same macro name and arguments, so a driver's CMakeLists keeps its VSTARGET lines; the tool behind it is
# tools/brkscatter, which REWRITES brick(...) / tile(...) lines into calls of the hand-written device code instead of generating
# code.  Written fresh; only the interface is the reference's.
find_package(Python 3.6 REQUIRED COMPONENTS Interpreter)
if (NOT DEFINED BRK_ROOT)
    get_filename_component(BRK_ROOT "${CMAKE_CURRENT_LIST_DIR}/.." ABSOLUTE)
endif ()
set(BrkScatter_SCRIPT ${BRK_ROOT}/tools/brkscatter)
set(VS_PREPROCESSOR cpp CACHE STRING "Preprocessor used by the rewrite pass")

macro(VSTARGET Name Input Output)
    set(_vs_inc "-I${BRK_ROOT}/include")
    get_property(_vs_dirs DIRECTORY ${CMAKE_CURRENT_SOURCE_DIR} PROPERTY INCLUDE_DIRECTORIES)
    foreach (_d ${_vs_dirs} ${${Name}_INCLUDE_DIRS})
        list(APPEND _vs_inc -I${_d})
    endforeach ()
    if (IS_ABSOLUTE "${Output}")
        set(VSTARGET_${Name}_OUTPUT "${Output}")
    else ()
        # the reference writes next to the source; generated files belong to the build tree here
        set(VSTARGET_${Name}_OUTPUT "${CMAKE_CURRENT_BINARY_DIR}/${Output}")
    endif ()
    if (IS_ABSOLUTE "${Input}")
        set(_vs_in "${Input}")
    else ()
        set(_vs_in "${CMAKE_CURRENT_SOURCE_DIR}/${Input}")
    endif ()
    add_custom_command(OUTPUT "${VSTARGET_${Name}_OUTPUT}"
            COMMAND ${CMAKE_COMMAND} -E env VSCPP=${VS_PREPROCESSOR} ${Python_EXECUTABLE} ${BrkScatter_SCRIPT} "${_vs_in}"
                    "${VSTARGET_${Name}_OUTPUT}" -- ${${Name}_COMPILE_OPTIONS} ${_vs_inc} -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -D__HIP__
            VERBATIM
            MAIN_DEPENDENCY "${_vs_in}"
            DEPENDS ${BrkScatter_SCRIPT} ${BRK_ROOT}/include/vecscatter.h
            COMMENT "[VS][${Name}] brick(...) / tile(...) rewrite"
            WORKING_DIRECTORY ${CMAKE_CURRENT_SOURCE_DIR})
endmacro()
