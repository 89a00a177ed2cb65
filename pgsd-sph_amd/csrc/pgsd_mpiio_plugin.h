/* pgsd_mpiio_plugin.h -- the interface between libpgsd_amd.so and its optional MPI-IO back end.
 *
 * libpgsd_amd.so does not link MPI (an MPI installation is not part of the GPU image's runtime, and an MPI's
 * handles and constants are not binary compatible across implementations).  The MPI-IO back end therefore is a
 * separate shared object, libpgsd_amd_mpiio.so, compiled against the MPI of the installation it runs with
 * (`make -C pgsd-sph_amd/csrc mpiio`), and loaded at run time when PGSD_IO=mpiio is set: every byte the library
 * writes or reads then goes through MPI_File_write_at / MPI_File_read_at at the offsets the POSIX back end uses --
 * the calls the reference makes (pgsd.c:2229, 1154, 2032, 1289-1306 / 651, 1501, 1559, 2534). */
#ifndef PGSD_MPIIO_PLUGIN_H
#define PGSD_MPIIO_PLUGIN_H

#ifdef __cplusplus
extern "C"
    {
#endif

    struct pgsd_mpiio_api
        {
        /* MPI_File_open on MPI_COMM_SELF (every rank opens the file for itself: the library's ranks need not be the
           ranks of one MPI communicator the plugin could know -- they may come through callbacks); 0 or -1 */
        int (*open)(const char* path, int readonly, void** fh);
        int (*close)(void* fh);
        /* bytes moved (short at the end of the file for a read) or -1; pieces of at most 1 GiB (MPI counts are int) */
        long long (*write_at)(void* fh, long long offset, const void* buf, long long bytes);
        long long (*read_at)(void* fh, long long offset, void* buf, long long bytes);
        int (*set_size)(void* fh, long long size);
        long long (*get_size)(void* fh); /* -1 on failure */
        const char* (*last_error)(void);
        };

    /* exported by libpgsd_amd_mpiio.so; fails (-1) when MPI is not initialised by the caller (the reference never
       calls MPI_Init either) */
    int pgsd_mpiio_plugin(struct pgsd_mpiio_api* out);

#ifdef __cplusplus
    }
#endif

#endif
