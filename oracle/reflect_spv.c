/* reflect_spv.c — TEST INFRASTRUCTURE (oracle/): prints what the reference's committed SPIR-V binaries say about the
 * DATA ABI of the hot path, as JSON, through the reference's own vendored reflection library.
 *
 * Linked against /root/reference/thirdparty/spirv-reflect/spirv_reflect.c, compiled where it lies by oracle/Makefile
 * (target _ref/reflect_spv; no reference source is copied into this repository). The engine itself validates its C++
 * push-constant structs against exactly this reflection at pipeline creation (deferred.cpp:30-62, pipelines.cpp:609-624,
 * shaders.cpp:537-598), so the numbers printed here are the layout the reference's shaders and host code agree on.
 *
 * This pins LAYOUT only — push-constant member offsets and sizes, the layouts of the buffer-reference structs, workgroup
 * size, image formats, descriptor set / binding numbers. It pins no arithmetic.
 *
 * usage: reflect_spv file.spv [file.spv ...]   -> one JSON object {"<basename>": {...}, ...} on stdout
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "spirv_reflect.h"

static void print_string(const char* s)
{
    putchar('"');
    for (; s && *s; s++)
    {
        if (*s == '"' || *s == '\\')
        {
            putchar('\\');
        }
        putchar(*s);
    }
    putchar('"');
}

static const char* type_name(const SpvReflectTypeDescription* t)
{
    if (!t)
    {
        return "";
    }
    return t->type_name ? t->type_name : "";
}

/* One block member: name, offsets, sizes, numeric shape, array shape, and (recursively) the members of a struct or of the
 * struct behind a buffer reference. `depth` bounds self-referential pointer types. */
static void print_member(const SpvReflectBlockVariable* v, int depth)
{
    printf("{\"name\": ");
    print_string(v->name);
    printf(", \"offset\": %u, \"absolute_offset\": %u, \"size\": %u, \"padded_size\": %u", v->offset, v->absolute_offset,
           v->size, v->padded_size);
    printf(", \"type_name\": ");
    print_string(type_name(v->type_description));
    printf(", \"type_flags\": %u", v->type_description ? (unsigned)v->type_description->type_flags : 0u);
    printf(", \"scalar_width\": %u, \"vector\": %u, \"columns\": %u, \"rows\": %u, \"matrix_stride\": %u",
           v->numeric.scalar.width, v->numeric.vector.component_count, v->numeric.matrix.column_count,
           v->numeric.matrix.row_count, v->numeric.matrix.stride);
    printf(", \"array_dims\": [");
    for (uint32_t i = 0; i < v->array.dims_count; i++)
    {
        printf("%s%u", i ? ", " : "", v->array.dims[i]);
    }
    printf("], \"array_stride\": %u, \"type_array_stride\": %u", v->array.stride,
           v->type_description ? v->type_description->traits.array.stride : 0u);
    printf(", \"members\": [");
    if (depth < 6)
    {
        for (uint32_t i = 0; i < v->member_count; i++)
        {
            if (i)
            {
                printf(", ");
            }
            print_member(&v->members[i], depth + 1);
        }
    }
    printf("]}");
}

static int reflect_file(const char* path)
{
    FILE* f = fopen(path, "rb");
    if (!f)
    {
        fprintf(stderr, "reflect_spv: cannot open %s\n", path);
        return 1;
    }
    fseek(f, 0, SEEK_END);
    long size = ftell(f);
    fseek(f, 0, SEEK_SET);
    void* code = malloc((size_t)size);
    if (!code || fread(code, 1, (size_t)size, f) != (size_t)size)
    {
        fprintf(stderr, "reflect_spv: cannot read %s\n", path);
        fclose(f);
        free(code);
        return 1;
    }
    fclose(f);

    SpvReflectShaderModule module;
    SpvReflectResult r = spvReflectCreateShaderModule((size_t)size, code, &module);
    if (r != SPV_REFLECT_RESULT_SUCCESS)
    {
        fprintf(stderr, "reflect_spv: %s: spvReflectCreateShaderModule failed (%d)\n", path, (int)r);
        free(code);
        return 1;
    }

    const char* base = strrchr(path, '/');
    base = base ? base + 1 : path;
    print_string(base);
    printf(": {\"bytes\": %ld, \"entry_point\": ", size);
    print_string(module.entry_point_name);
    printf(", \"stage\": %u", (unsigned)module.shader_stage);
    {
        const SpvReflectEntryPoint* e = spvReflectGetEntryPoint(&module, module.entry_point_name);
        printf(", \"local_size\": [%u, %u, %u]", e ? e->local_size.x : 0, e ? e->local_size.y : 0, e ? e->local_size.z : 0);
    }

    uint32_t count = 0;
    spvReflectEnumeratePushConstantBlocks(&module, &count, NULL);
    SpvReflectBlockVariable** blocks = (SpvReflectBlockVariable**)calloc(count ? count : 1, sizeof(*blocks));
    spvReflectEnumeratePushConstantBlocks(&module, &count, blocks);
    printf(", \"push_constants\": [");
    for (uint32_t i = 0; i < count; i++)
    {
        if (i)
        {
            printf(", ");
        }
        print_member(blocks[i], 0);
    }
    printf("]");
    free(blocks);

    count = 0;
    spvReflectEnumerateDescriptorBindings(&module, &count, NULL);
    SpvReflectDescriptorBinding** bindings = (SpvReflectDescriptorBinding**)calloc(count ? count : 1, sizeof(*bindings));
    spvReflectEnumerateDescriptorBindings(&module, &count, bindings);
    printf(", \"bindings\": [");
    for (uint32_t i = 0; i < count; i++)
    {
        const SpvReflectDescriptorBinding* b = bindings[i];
        printf("%s{\"name\": ", i ? ", " : "");
        print_string(b->name);
        printf(", \"set\": %u, \"binding\": %u, \"descriptor_type\": %u, \"count\": %u, \"image_dim\": %u, \"image_format\": %u, "
               "\"image_sampled\": %u, \"image_depth\": %u}",
               b->set, b->binding, (unsigned)b->descriptor_type, b->count, (unsigned)b->image.dim, (unsigned)b->image.image_format,
               b->image.sampled, b->image.depth);
    }
    printf("]}");
    free(bindings);

    spvReflectDestroyShaderModule(&module);
    free(code);
    return 0;
}

int main(int argc, char** argv)
{
    if (argc < 2)
    {
        fprintf(stderr, "usage: %s file.spv [...]\n", argv[0]);
        return 2;
    }
    printf("{");
    for (int i = 1; i < argc; i++)
    {
        if (i > 1)
        {
            printf(",\n ");
        }
        if (reflect_file(argv[i]))
        {
            return 1;
        }
    }
    printf("}\n");
    return 0;
}
