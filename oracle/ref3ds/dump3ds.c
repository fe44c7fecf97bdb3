/* Test tooling, not product code.  Runs the REAL lib3ds 1.3.0 (compiled from /root/reference/lib3ds-1.3.0 where it lies,
 * by oracle/ref3ds/Makefile, into oracle/_ref/) the way the reference's loader drives it (src/Loader.cc:276-353:
 * lib3ds_file_load, lib3ds_file_eval(0), lib3ds_mesh_calculate_normals per mesh, materials by name, diffuse colour to
 * bytes, two_sided) and writes what that code pushes into Scene::_vertices/_triangles BEFORE the loader's common tail:
 *   "R3DS" u32 n_tri, then per triangle 3 x (pos[3] f32, normal[3] f32), u32 r, g, b, u32 two_sided.
 * tests/golden/legocar_3ds.r3ds.xz was made with it (scripts/make_3ds_golden.sh); renderer_amd's own .3ds reader
 * must reproduce that file bit for bit, and the oracle renders from it. */
#include <lib3ds/file.h>
#include <lib3ds/material.h>
#include <lib3ds/mesh.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static void w32(FILE *f, uint32_t v) { fwrite(&v, 4, 1, f); }
static void wf(FILE *f, float v) { fwrite(&v, 4, 1, f); }

int main(int argc, char **argv)
{
    if (argc != 3) { fprintf(stderr, "usage: dump3ds in.3ds out.r3ds\n"); return 2; }
    Lib3dsFile *file = lib3ds_file_load(argv[1]);
    if (!file) { fprintf(stderr, "lib3ds could not load %s\n", argv[1]); return 1; }
    lib3ds_file_eval(file, 0);
    uint32_t total = 0, meshes = 0, flipped = 0;
    for (Lib3dsMesh *m = file->meshes; m; m = m->next)
        if (m->pointL && m->faceL) { total += m->faces; meshes++; }
    FILE *out = fopen(argv[2], "wb");
    if (!out) return 1;
    fwrite("R3DS", 1, 4, out);
    w32(out, total);
    for (Lib3dsMesh *m = file->meshes; m; m = m->next) {
        if (!m->pointL || !m->faceL) continue;
        if (lib3ds_matrix_det(m->matrix) < 0.0) flipped++;
        Lib3dsVector *normals = malloc(sizeof(Lib3dsVector) * 3 * m->faces);
        lib3ds_mesh_calculate_normals(m, normals);
        for (unsigned i = 0; i < m->faces; i++) {
            Lib3dsMaterial *mat = NULL;
            for (Lib3dsMaterial *q = file->materials; q; q = q->next)       /* std::map keeps the first of equal names */
                if (!strcmp(q->name, m->faceL[i].material)) { mat = q; break; }
            uint32_t r = 255, g = 255, b = 255, two = 0;
            if (mat) {
                r = (unsigned)(255.0 * mat->diffuse[0]);
                g = (unsigned)(255.0 * mat->diffuse[1]);
                b = (unsigned)(255.0 * mat->diffuse[2]);
                two = mat->two_sided != 0;
            }
            for (int k = 0; k < 3; k++) {
                const unsigned p = m->faceL[i].points[k];
                for (int a = 0; a < 3; a++) wf(out, m->pointL[p].pos[a]);
                for (int a = 0; a < 3; a++) wf(out, normals[3 * i + k][a]);
            }
            w32(out, r); w32(out, g); w32(out, b); w32(out, two);
        }
        free(normals);
    }
    fclose(out);
    fprintf(stderr, "%u meshes, %u triangles, %u meshes with a mirrored matrix\n", meshes, total, flipped);
    return 0;
}
